// Host-side construction of the front-end tables and of the packed weight image.  Pure CPU code
// (double precision, rounded once to float32): callable without a GPU through the C ABI so the
// CPU test-suite can check it against the oracle.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ww_internal.h"

namespace ww {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }

// ---- Slaney mel scale: librosa.hz_to_mel / mel_to_hz with htk=False ---------------------------
static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(sr=16000, n_fft=2048, n_mels=80, fmin=0, fmax=8000, norm='slaney', dtype=float32):
// triangle stored as float32, then scaled by the float64 area normaliser and rounded again.
void build_mel_filterbank(float* out) {
    const int n_edges = kMels + 2;
    std::vector<double> edges(n_edges);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(8000.0);
    const double step = (m1 - m0) / (n_edges - 1);          // np.linspace: start + i*step, last = stop
    for (int i = 0; i < n_edges; ++i) edges[i] = mel_to_hz(i == n_edges - 1 ? m1 : m0 + i * step);
    for (int i = 0; i < kMels; ++i) {
        const double fd0 = edges[i + 1] - edges[i], fd1 = edges[i + 2] - edges[i + 1];
        const double enorm = 2.0 / (edges[i + 2] - edges[i]);
        for (int k = 0; k < kBins; ++k) {
            const double f = k * (double(WW_SAMPLE_RATE) / kNfft);
            const double lower = -(edges[i] - f) / fd0, upper = (edges[i + 2] - f) / fd1;
            const double tri = std::fmax(0.0, std::fmin(lower, upper));
            out[i * kBins + k] = float(double(float(tri)) * enorm);
        }
    }
}

void build_hann(double* w) {
    for (int n = 0; n < kNfft; ++n) w[n] = 0.5 - 0.5 * std::cos(2.0 * M_PI * n / kNfft);
}

static float2 twiddle(int N, long e) {   // W_N^e = exp(-2*pi*i*e/N), argument reduced exactly
    e %= N;
    const double a = -2.0 * M_PI * double(e) / double(N);
    return make_float2(float(std::cos(a)), float(std::sin(a)));
}

int build_logmel_tables(LogmelTables* t) {
    std::memset(t, 0, sizeof(*t));
    std::vector<double> w(kNfft);
    build_hann(w.data());
    for (int n = 0; n < kNfft; ++n) t->window[n] = float(w[n]);
    for (int k1 = 1; k1 < 8; ++k1)
        for (int n = 0; n < 128; ++n) t->tw1[k1 - 1][n] = twiddle(1024, long(n) * k1);
    for (int k2 = 1; k2 < 8; ++k2)
        for (int n = 0; n < 16; ++n) t->tw2[k2 - 1][n] = twiddle(128, long(n) * k2);
    for (int k = 0; k < 512; ++k) t->twp[k] = twiddle(2048, k == 0 ? 512 : k);

    std::vector<float> M(size_t(kMels) * kBins);
    build_mel_filterbank(M.data());
    int p = 0;
    for (int f = 0; f < kMels; ++f) {
        int lo = -1, hi = -1;
        for (int k = 0; k < kBins; ++k)
            if (M[f * kBins + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
        if (lo < 1 || hi > 1023) return fail(WW_EINVAL, "mel filter %d touches bin 0 or 1024", f);
        for (int k = lo; k <= hi; ++k)   // support must be one contiguous run
            if (M[f * kBins + k] == 0.0f) return fail(WW_EINVAL, "mel filter %d has a hole at bin %d", f, k);
        t->filt_p0[f] = p;
        for (int k0 = lo; k0 <= hi; k0 += kPieceLen, ++p) {
            if (p >= kPieces) return fail(WW_EINVAL, "more than %d mel pieces", kPieces);
            // keep the 8-bin read inside bins 1..1023 (zero weights past the filter's end)
            const int start = k0 + kPieceLen - 1 > 1023 ? 1023 - (kPieceLen - 1) : k0;
            t->piece_k0[p] = start;
            for (int i = 0; i < kPieceLen; ++i) {
                const int k = start + i;
                t->piece_w[p][i] = (k >= k0 && k <= hi && k < k0 + kPieceLen) ? M[f * kBins + k] : 0.0f;
            }
        }
        t->filt_cnt[f] = p - t->filt_p0[f];
    }
    const int real = p;
    for (; p < kPieces; ++p) t->piece_k0[p] = 1;   // padding pieces: weight 0 on valid bins 1..8
    return real;
}

// ---- packed weights --------------------------------------------------------------------------
PackedLayout packed_layout(int n_conv) {
    PackedLayout L{};
    L.n_conv = n_conv;
    L.c_last = n_conv == 3 ? 128 : 64;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += (n + 3) & ~int64_t(3); return at; };   // 16-byte aligned blocks
    L.conv1_w = take(32 * 9);
    L.conv1_b = take(32);
    L.conv2_w = take(2 * 144 * 64);
    L.conv2_b = take(64);
    L.conv3_w = n_conv == 3 ? take(4 * 288 * 64) : -1;
    L.conv3_b = n_conv == 3 ? take(128) : -1;
    L.l0_w = take(int64_t(L.c_last) * kGateCols);
    L.l0_b = take(kGateCols);
    L.l1_w = take(int64_t(kHidden) * kGateCols);
    L.l1_b = take(kGateCols);
    L.fc_w = take(2 * kHidden);
    L.fc_b = take(4);
    L.total = o;
    return L;
}

// conv weight [Cout][Cin][3][3] (torch) -> MFMA 32x32x2 B-operand order:
// out[(nt*KS + (c*3+dy)*3+dx)*64 + lane] = W[32*nt + (lane&31)][2*c + (lane>>5)][dy][dx],  KS = Cin/2*9
static void pack_conv_b_operand(const float* w, int cout, int cin, float* out) {
    const int ks = cin / 2 * 9;
    for (int nt = 0; nt < cout / 32; ++nt)
        for (int c = 0; c < cin / 2; ++c)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = 32 * nt + (lane & 31), ci = 2 * c + (lane >> 5);
                        out[(int64_t(nt) * ks + (c * 3 + dy) * 3 + dx) * 64 + lane] = w[((co * cin + ci) * 3 + dy) * 3 + dx];
                    }
}

// weight_ih [4H][K] (gate rows i,f,g,o) -> [K][768], column (hb*3 + gate)*32 + u  <-  row goff[gate] + 32*hb + u
static void pack_lstm(const float* w_ih, const float* b_ih, const float* b_hh, int K, float* wt, float* b) {
    const int goff[3] = {0, 2 * kHidden, 3 * kHidden};   // i, g, o
    for (int hb = 0; hb < kHidden / 32; ++hb)
        for (int g = 0; g < 3; ++g)
            for (int u = 0; u < 32; ++u) {
                const int col = (hb * 3 + g) * 32 + u, row = goff[g] + 32 * hb + u;
                for (int k = 0; k < K; ++k) wt[int64_t(k) * kGateCols + col] = w_ih[int64_t(row) * K + k];
                b[col] = b_ih[row] + b_hh[row];
            }
}

}  // namespace ww

using namespace ww;

extern "C" {

int ww_abi_version(void) { return WW_ABI_VERSION; }
const char* ww_last_error(void) { return ww::last_error(); }

int ww_mel_filterbank_host(float* out_host) {
    if (!out_host) return fail(WW_EINVAL, "null output");
    build_mel_filterbank(out_host);
    return WW_OK;
}

int ww_hann_window_host(float* out_host) {
    if (!out_host) return fail(WW_EINVAL, "null output");
    std::vector<double> w(kNfft);
    build_hann(w.data());
    for (int n = 0; n < kNfft; ++n) out_host[n] = float(w[n]);
    return WW_OK;
}

int64_t ww_packed_weights_floats(int32_t n_conv) {
    if (n_conv != 2 && n_conv != 3) return fail(WW_EINVAL, "n_conv must be 2 or 3, got %d", n_conv);
    return packed_layout(n_conv).total;
}

int ww_pack_weights_host(const ww_state_dict* sd, float* out) {
    if (!sd || !out) return fail(WW_EINVAL, "null argument");
    if (sd->n_conv != 2 && sd->n_conv != 3) return fail(WW_EINVAL, "n_conv must be 2 or 3, got %d", sd->n_conv);
    if (sd->hidden != kHidden) return fail(WW_EUNSUPPORTED, "hidden size %d (only %d is built)", sd->hidden, kHidden);
    for (int i = 0; i < sd->n_conv; ++i)
        if (!sd->conv_weight[i] || !sd->conv_bias[i]) return fail(WW_EINVAL, "conv%d weight/bias missing", i + 1);
    for (int l = 0; l < 2; ++l)
        if (!sd->lstm_weight_ih[l] || !sd->lstm_bias_ih[l] || !sd->lstm_bias_hh[l])
            return fail(WW_EINVAL, "lstm layer %d weight/bias missing", l);
    if (!sd->fc_weight || !sd->fc_bias) return fail(WW_EINVAL, "fc weight/bias missing");

    const PackedLayout L = packed_layout(sd->n_conv);
    std::memset(out, 0, sizeof(float) * L.total);
    std::memcpy(out + L.conv1_w, sd->conv_weight[0], sizeof(float) * 32 * 9);
    std::memcpy(out + L.conv1_b, sd->conv_bias[0], sizeof(float) * 32);
    pack_conv_b_operand(sd->conv_weight[1], 64, 32, out + L.conv2_w);
    std::memcpy(out + L.conv2_b, sd->conv_bias[1], sizeof(float) * 64);
    if (sd->n_conv == 3) {
        pack_conv_b_operand(sd->conv_weight[2], 128, 64, out + L.conv3_w);
        std::memcpy(out + L.conv3_b, sd->conv_bias[2], sizeof(float) * 128);
    }
    pack_lstm(sd->lstm_weight_ih[0], sd->lstm_bias_ih[0], sd->lstm_bias_hh[0], L.c_last, out + L.l0_w, out + L.l0_b);
    pack_lstm(sd->lstm_weight_ih[1], sd->lstm_bias_ih[1], sd->lstm_bias_hh[1], kHidden, out + L.l1_w, out + L.l1_b);
    std::memcpy(out + L.fc_w, sd->fc_weight, sizeof(float) * 2 * kHidden);
    std::memcpy(out + L.fc_b, sd->fc_bias, sizeof(float) * 2);
    return WW_OK;
}

}  // extern "C"
