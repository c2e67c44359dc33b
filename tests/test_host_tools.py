"""The small host-side tools the measurements lean on: they have to keep reading what the compiler and the kernel write."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ASM = """
\t.text
_ZN2ww5demo1Ev:                          ; @_ZN2ww5demo1Ev
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tscratch_store_dword off, v1, off
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]
\tscratch_load_dword v1, off, off
\ts_waitcnt vmcnt(0)
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
_ZN2ww5demo2Ev:                          ; @_ZN2ww5demo2Ev
\tv_mov_b32_e32 v0, 0
\ts_endpgm
"""


def test_isa_scratch_names_the_reload_inside_the_loop(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(ASM)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_scratch.py"), str(p)], capture_output=True, text=True, check=True).stdout
    lines = out.strip().splitlines()
    assert lines[0].startswith("_ZN2ww5demo1Ev") and "1 MFMAs" in lines[0] and "2 scratch instructions" in lines[0]
    assert "scratch_store_dword" in lines[1] and "(no loop)" in lines[1]
    assert "scratch_load_dword" in lines[2] and "innermost loop" in lines[2] and "(1 MFMAs in it)" in lines[2]
    assert "demo2" not in out                                     # kernels without scratch are not listed


def test_cgroup_cpu_stat_is_a_dict_of_ints_or_empty():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_files
    st = bench_files._cgroup_cpu_stat()
    assert isinstance(st, dict) and all(isinstance(v, int) for v in st.values())
    if st:
        assert "usage_usec" in st
