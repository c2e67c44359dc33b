// Wave-level 1024-point complex FFT for the augmentation kernels (STFT / inverse STFT of the phase vocoder).
// Same decomposition as K1 (ww_logmel.hip): radix 8 x 8 x 16, 16 points per lane, two exchanges through an 8 KiB LDS
// slab that belongs to the wave (LDS is in-order per wave: no workgroup barrier), XOR-swizzled so that every
// ds_read/write_b128 is bank-conflict free.  K1 keeps its own hand-scheduled copy of these passes.
#pragma once
#include "ww_internal.h"

namespace ww {
namespace fft {

constexpr int kSlabFloats = 2048 + 4;   // 1024 complex (+4: 16-byte aligned, staggers consecutive slabs over the banks)

__device__ __forceinline__ float2 add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// forward DFTs (e^{-2 pi i nk/N}), natural order in and out, all indices static -> registers
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = add(a0, a2), t1 = sub(a0, a2), t2 = add(a1, a3), t3 = mul_neg_i(sub(a1, a3));
    a0 = add(t0, t2); a2 = sub(t0, t2); a1 = add(t1, t3); a3 = sub(t1, t3);
}

__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float c = 0.70710678118654752440f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    o1 = make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));      // * W8^1
    o2 = mul_neg_i(o2);                                           // * W8^2
    o3 = make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));      // * W8^3
    v[0] = add(e0, o0); v[4] = sub(e0, o0);
    v[1] = add(e1, o1); v[5] = sub(e1, o1);
    v[2] = add(e2, o2); v[6] = sub(e2, o2);
    v[3] = add(e3, o3); v[7] = sub(e3, o3);
}

__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    float2 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    dft8(e);
    dft8(o);
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    constexpr float c2 = 0.70710678118654752440f;
    const float2 w[8] = {{1.f, 0.f}, {c1, -s1}, {c2, -c2}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-c2, -c2}, {-c1, -s1}};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 t = (k == 0) ? o[0] : (k == 4 ? mul_neg_i(o[4]) : cmul(o[k], w[k]));
        v[k] = add(e[k], t);
        v[k + 8] = sub(e[k], t);
    }
}

__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// float2 index of Z[k] inside the slab after wave_fft1024
__device__ __forceinline__ int zpos(int k) { return k ^ (((k >> 4) & 3) << 1); }

// In : lane holds z[128 n1 + 2 lane] in za[n1] and z[128 n1 + 2 lane + 1] in zb[n1], n1 = 0..7.
// Out: Z[k] = sum_n z[n] e^{-2 pi i nk/1024} at float2 index zpos(k) of the wave's slab (valid after the call).
__device__ __forceinline__ void wave_fft1024(float2 (&za)[8], float2 (&zb)[8], float* slab, const LogmelTables* tb, int lane) {
    float4* slab4 = reinterpret_cast<float4*>(slab);
    float2* slab2 = reinterpret_cast<float2*>(slab);
    const float4* tw1_4 = reinterpret_cast<const float4*>(&tb->tw1[0][0]);   // [7][64]: twiddles of n' = 2 lane, 2 lane + 1
    const float4* tw2_4 = reinterpret_cast<const float4*>(&tb->tw2[0][0]);   // [7][8]
    // pass 1: radix 8 over n1 (stride 128), twiddle W_1024^{n' k1}.  Twiddles are fetched as a group ahead of the
    // butterflies (written at their use each load is issued right there: one L1 round trip per twiddle).
    float4 t1[7];
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) t1[k1 - 1] = tw1_4[(k1 - 1) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    dft8(za);
    dft8(zb);
    slab4[lane] = make_float4(za[0].x, za[0].y, zb[0].x, zb[0].y);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) {
        const float4 t = t1[k1 - 1];
        const float2 a = cmul(za[k1], make_float2(t.x, t.y));
        const float2 b = cmul(zb[k1], make_float2(t.z, t.w));
        slab4[k1 * 64 + (lane ^ (8 * ((k1 >> 1) & 1)))] = make_float4(a.x, a.y, b.x, b.y);
    }
    lds_order();
    // pass 2: lane = (k1, j): radix 8 over n2 of y[k1][16 n2 + 2j + q], twiddle W_128
    {
        const int k1r = lane >> 3, jr = lane & 7;
        const int s8 = 8 * ((k1r >> 1) & 1);
        const float4* x1e = slab4 + k1r * 64 + jr + s8;
        const float4* x1o = slab4 + k1r * 64 + jr - s8;
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const float4 v = (n2 & 1) ? x1o[n2 * 8] : x1e[n2 * 8];
            za[n2] = make_float2(v.x, v.y);
            zb[n2] = make_float2(v.z, v.w);
        }
        float4 t2[7];
#pragma unroll
        for (int k2 = 1; k2 < 8; ++k2) t2[k2 - 1] = tw2_4[(k2 - 1) * 8 + jr];
        lds_order();
        __builtin_amdgcn_sched_barrier(0);
        dft8(za);
        dft8(zb);
        float4* x2w[4];
#pragma unroll
        for (int hk = 0; hk < 4; ++hk) x2w[hk] = slab4 + 64 * k1r + (jr ^ (4 * (k1r & 1) + hk));
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            float2 a = za[k2], b = zb[k2];
            if (k2 > 0) {
                const float4 t = t2[k2 - 1];
                a = cmul(a, make_float2(t.x, t.y));
                b = cmul(b, make_float2(t.z, t.w));
            }
            x2w[k2 >> 1][8 * k2] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
    lds_order();
    // pass 3: lane = (k1, k2): radix 16 over n'' -> Z[k1 + 8 k2 + 64 k'']
    {
        float2 u[16];
        const int sw2 = (lane >> 1) & 7;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float4 v = slab4[lane * 8 + (m ^ sw2)];
            u[2 * m] = make_float2(v.x, v.y);
            u[2 * m + 1] = make_float2(v.z, v.w);
        }
        lds_order();
        dft16(u);
        const int lp = (lane >> 3) + 8 * (lane & 7);
        float2* zw = slab2 + zpos(lp);                       // bits 4-5 of k = lp + 64 kk are lp's: one base
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) zw[64 * kk] = u[kk];
    }
    lds_order();
}

}  // namespace fft
}  // namespace ww
