"""CPU oracle for the wakeword hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (`wakeword-jupyterlab_amd/`) imports this
directory.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it, and there only as the checker / the timed
CPU baseline -- never as the thing that is shipped or measured as the GPU path.

Contents
--------
mel_oracle.py    numpy restatement of `librosa.feature.melspectrogram` +
                 `librosa.power_to_db` as called from
                 /root/reference/wakeword_training_script.py:85-101.
                 librosa 0.10.1 (README.md:386) is an un-vendored third-party
                 dependency that is not installed in the build image, and the
                 reference holds no golden vectors for it: **mel parity is
                 UNPINNED** against librosa itself.  The restatement is pinned
                 instead by (i) the published algorithm, (ii) an independent
                 `torch.stft` cross-check of the STFT stage and (iii) the
                 structural properties recorded in SURVEY.md section 8(c).
model_oracle.py  closed-form restatement of `SimpleWakewordModel.forward`
                 (/root/reference/wakeword_training/train_wakeword.py:38-49) and
                 `WakewordModel.forward` (wakeword_training_script.py:167-184).
                 PINNED: tests/golden/model_simple_*.npz were produced by
                 importing the reference module itself (tests/golden/make_golden.py).
decode_oracle.py restatement of load_audio's numeric part + normalise + crop/pad
                 (wakeword_training_script.py:65-83,125-133); resampler = scipy's polyphase design,
                 UNPINNED against librosa's soxr_hq.
augment_oracle.py restatement of augment_audio (wakeword_training_script.py:103-123): librosa 0.10.1's
                 stft / phase_vocoder / istft, resampy's kaiser_best as the resampler stand-in, the
                 build's hash generator as the noise stand-in.  UNPINNED against librosa.
"""
