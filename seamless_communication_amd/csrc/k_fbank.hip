// 80-bin log-mel filterbank front-end (kaldi-compatible), one workgroup per
// 25 ms frame: LDS window -> DC removal -> pre-emphasis -> Povey window ->
// 512-point radix-2 FFT in LDS -> power spectrum -> mel matvec -> log.
// Restates reference ggml/examples/kaldi-native-fbank/csrc/feature-window.cc:121-233,
// feature-functions.cc:28-47, mel-computations.cc:224-247, feature-fbank.cc:73-118.
#include "kernels.h"

namespace sc {

static constexpr int FRAME_LEN = 400;
static constexpr int FRAME_SHIFT = 160;
static constexpr int NFFT = 512;
static constexpr int NBINS = 80;

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_red[wave] = v;
    __syncthreads();
    const float r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return r;
}

// consts layout: window[400] | melT[256][80] | tw_cos[256] | tw_sin[256]
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wav, int64_t wav_stride,
                                                    const int* __restrict__ num_samples,
                                                    float* __restrict__ out, int t_rows,
                                                    const float* __restrict__ consts, float scale) {
    __shared__ float s_x[NFFT];
    __shared__ float s_re[NFFT];
    __shared__ float s_im[NFFT];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const int n = blockIdx.y;
    const int ns = num_samples[n];
    const int nframes = ns < FRAME_LEN ? 0 : 1 + (ns - FRAME_LEN) / FRAME_SHIFT;
    float* orow = out + ((int64_t)n * t_rows + f) * NBINS;
    if (f >= nframes) {  // padding rows (Collater pad_value = 0)
        if (tid < NBINS) orow[tid] = 0.f;
        return;
    }
    const float* w = wav + (int64_t)n * wav_stride + (int64_t)f * FRAME_SHIFT;
    const float* window = consts;
    const float* melT = consts + FRAME_LEN;
    const float* tw_cos = melT + 256 * NBINS;
    const float* tw_sin = tw_cos + 256;

    // load + scale
    float part = 0.f;
    for (int i = tid; i < NFFT; i += 256) {
        const float v = i < FRAME_LEN ? w[i] * scale : 0.f;
        s_x[i] = v;
        part += v;
    }
    __syncthreads();
    const float mean = block_sum_256(part, s_red) / (float)FRAME_LEN;
    // DC removal, pre-emphasis (0.97), window; write bit-reversed for the DIT FFT
    for (int i = tid; i < NFFT; i += 256) {
        float v = 0.f;
        if (i < FRAME_LEN) {
            const float cur = s_x[i] - mean;
            const float prev = (i > 0 ? s_x[i - 1] : s_x[0]) - mean;
            v = (cur - 0.97f * prev) * window[i];
        }
        const int rev = __brev((unsigned)i) >> (32 - 9);
        s_re[rev] = v;
        s_im[rev] = 0.f;
    }
    __syncthreads();
    // 9 radix-2 DIT stages, one butterfly per thread per stage
    for (int stage = 0; stage < 9; ++stage) {
        const int half = 1 << stage;
        const int j = tid & (half - 1);
        const int i0 = ((tid >> stage) << (stage + 1)) + j;
        const int i1 = i0 + half;
        const int tw = j << (8 - stage);  // index into the 256-entry table: exp(-2*pi*i*tw/512)
        const float c = tw_cos[tw], sn = tw_sin[tw];
        const float xr = s_re[i1], xi = s_im[i1];
        const float tr = xr * c - xi * sn;
        const float ti = xr * sn + xi * c;
        const float ur = s_re[i0], ui = s_im[i0];
        s_re[i0] = ur + tr;
        s_im[i0] = ui + ti;
        s_re[i1] = ur - tr;
        s_im[i1] = ui - ti;
        __syncthreads();
    }
    // power spectrum of bins 0..255 (bin 256 is never used by the mel banks)
    {
        const float re = s_re[tid], im = s_im[tid];
        s_x[tid] = re * re + im * im;
    }
    __syncthreads();
    if (tid < NBINS) {
        float e = 0.f;
        for (int k = 0; k < 256; ++k) e = fmaf(melT[k * NBINS + tid], s_x[k], e);
        // std::max(e, FLT_EPS) of feature-fbank.cc:105 keeps a NaN energy (corrupted audio must stay visible to
        // the caller's NaN filter, cli/m4t/evaluate/evaluate.py:278-289); fmaxf would replace it by the floor
        orow[tid] = logf(e < 1.1920928955078125e-07f ? 1.1920928955078125e-07f : e);
    }
}

void launch_fbank(const float* wav, int64_t wav_stride, const int* num_samples, int nb, float* out,
                  int t_rows, const float* consts, float scale, hipStream_t s) {
    if (nb <= 0 || t_rows <= 0) return;
    hipLaunchKernelGGL(fbank_kernel, dim3(t_rows, nb), dim3(256), 0, s, wav, wav_stride, num_samples, out,
                       t_rows, consts, scale);
    SC_LAUNCH_CHECK();
}

// The same front-end at ANY sample rate (fairseq2n hands the waveform's own rate to kaldi and does not resample,
// inference/translator.py:270-292): window = int(rate * 0.025) samples, shift = int(rate * 0.010), FFT size = the next power of
// two (feature-window.h FrameExtractionOptions), mel banks up to the rate's Nyquist.  One workgroup per frame, 256 threads,
// NFFT = 2^LOG2N in {256 .. 2048} points; consts layout: window[frame_len] | melT[NFFT/2][80] | tw_cos[NFFT/2] | tw_sin[NFFT/2].
// The 16 kHz kernel above is this kernel at LOG2N = 9 with its loops unrolled; it stays the path of the model's own rate.
template <int LOG2N>
__global__ __launch_bounds__(256) void fbank_any_kernel(const float* __restrict__ wav, int64_t wav_stride, const int* __restrict__ num_samples,
                                                        float* __restrict__ out, int t_rows, const float* __restrict__ consts, float scale,
                                                        int frame_len, int frame_shift) {
    constexpr int N = 1 << LOG2N, HALF = N / 2;
    __shared__ float s_x[N];
    __shared__ float s_re[N];
    __shared__ float s_im[N];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    const int n = blockIdx.y;
    const int ns = num_samples[n];
    const int nframes = ns < frame_len ? 0 : 1 + (ns - frame_len) / frame_shift;
    float* orow = out + ((int64_t)n * t_rows + f) * NBINS;
    if (f >= nframes) {
        if (tid < NBINS) orow[tid] = 0.f;
        return;
    }
    const float* w = wav + (int64_t)n * wav_stride + (int64_t)f * frame_shift;
    const float* window = consts;
    const float* melT = consts + frame_len;
    const float* tw_cos = melT + HALF * NBINS;
    const float* tw_sin = tw_cos + HALF;
    float part = 0.f;
    for (int i = tid; i < N; i += 256) {
        const float v = i < frame_len ? w[i] * scale : 0.f;
        s_x[i] = v;
        part += v;
    }
    __syncthreads();
    const float mean = block_sum_256(part, s_red) / (float)frame_len;
    for (int i = tid; i < N; i += 256) {
        float v = 0.f;
        if (i < frame_len) {
            const float cur = s_x[i] - mean;
            const float prev = (i > 0 ? s_x[i - 1] : s_x[0]) - mean;
            v = (cur - 0.97f * prev) * window[i];
        }
        const int rev = __brev((unsigned)i) >> (32 - LOG2N);
        s_re[rev] = v;
        s_im[rev] = 0.f;
    }
    __syncthreads();
    for (int stage = 0; stage < LOG2N; ++stage) {
        const int half = 1 << stage;
        for (int b = tid; b < HALF; b += 256) {  // butterfly b of this stage
            const int j = b & (half - 1);
            const int i0 = ((b >> stage) << (stage + 1)) + j;
            const int i1 = i0 + half;
            const int tw = j << (LOG2N - 1 - stage);  // exp(-2 pi i tw / N)
            const float c = tw_cos[tw], sn = tw_sin[tw];
            const float xr = s_re[i1], xi = s_im[i1];
            const float tr = xr * c - xi * sn;
            const float ti = xr * sn + xi * c;
            const float ur = s_re[i0], ui = s_im[i0];
            s_re[i0] = ur + tr;
            s_im[i0] = ui + ti;
            s_re[i1] = ur - tr;
            s_im[i1] = ui - ti;
        }
        __syncthreads();
    }
    for (int k = tid; k < HALF; k += 256) {
        const float re = s_re[k], im = s_im[k];
        s_x[k] = re * re + im * im;
    }
    __syncthreads();
    if (tid < NBINS) {
        float e = 0.f;
        for (int k = 0; k < HALF; ++k) e = fmaf(melT[k * NBINS + tid], s_x[k], e);
        orow[tid] = logf(e < 1.1920928955078125e-07f ? 1.1920928955078125e-07f : e);
    }
}

void launch_fbank_any(const float* wav, int64_t wav_stride, const int* num_samples, int nb, float* out, int t_rows, const float* consts,
                      float scale, int frame_len, int frame_shift, int nfft, hipStream_t s) {
    if (nb <= 0 || t_rows <= 0) return;
    SC_CHECK(frame_len >= 2 && frame_shift >= 1 && frame_len <= nfft, "fbank: window %d / shift %d / FFT %d", frame_len, frame_shift, nfft);
    const dim3 grid(t_rows, nb), block(256);
#define SC_FB(L) hipLaunchKernelGGL((fbank_any_kernel<L>), grid, block, 0, s, wav, wav_stride, num_samples, out, t_rows, consts, scale, frame_len, frame_shift)
    switch (nfft) {
        case 256: SC_FB(8); break;
        case 512: SC_FB(9); break;
        case 1024: SC_FB(10); break;
        case 2048: SC_FB(11); break;
        default: SC_CHECK(false, "fbank: FFT size %d (supported sample rates: 5.2 - 81.9 kHz, window of 129 .. 2048 samples)", nfft);
    }
#undef SC_FB
    SC_LAUNCH_CHECK();
}

// Per-utterance, per-bin standardisation over the valid frames: (x - mean) / std
// with the unbiased std and no epsilon (fairseq2n at::std_mean semantics).
// One workgroup = one utterance x 16 bins x 64 time slices (a thread strides over the frames of its bin); the
// per-slice double sums meet in LDS and are added in slice order by every thread of the bin, so the result does not
// depend on the launch shape.  (The first version walked the frames serially with one thread per bin: 0.5 ms per call.)
constexpr int STD_BINS = 16, STD_SLICES = 64;

__device__ __forceinline__ double std_slice_sum(double (*part)[STD_BINS], int cl, int ts, double v) {
    __syncthreads();  // the previous round's readers are done
    part[ts][cl] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < STD_SLICES; ++i) s += part[i][cl];
    return s;
}

__global__ __launch_bounds__(STD_BINS* STD_SLICES) void standardize_kernel(float* __restrict__ feat, int t_rows,
                                                                           const int* __restrict__ num_frames, int C) {
    __shared__ double part[STD_SLICES][STD_BINS];
    const int n = blockIdx.x;
    const int cl = threadIdx.x % STD_BINS, ts = threadIdx.x / STD_BINS;
    const int c = blockIdx.y * STD_BINS + cl;
    const bool live = c < C;
    const int T = num_frames[n];
    float* base = feat + (int64_t)n * t_rows * C + (live ? c : 0);
    double s = 0.0;
    if (live)
        for (int t = ts; t < T; t += STD_SLICES) s += (double)base[(int64_t)t * C];
    s = std_slice_sum(part, cl, ts, s);
    const double mean = T > 0 ? s / T : 0.0;
    double q = 0.0;
    if (live)
        for (int t = ts; t < T; t += STD_SLICES) {
            const double d = (double)base[(int64_t)t * C] - mean;
            q += d * d;
        }
    q = std_slice_sum(part, cl, ts, q);
    const double stdv = sqrt(q / (double)(T - 1));
    if (live)
        for (int t = ts; t < T; t += STD_SLICES) base[(int64_t)t * C] = (float)(((double)base[(int64_t)t * C] - mean) / stdv);
}

void launch_standardize(float* feat, int nb, int t_rows, const int* num_frames, int C, hipStream_t s) {
    if (nb <= 0 || C <= 0) return;
    hipLaunchKernelGGL(standardize_kernel, dim3(nb, (C + STD_BINS - 1) / STD_BINS), dim3(STD_BINS * STD_SLICES), 0, s, feat, t_rows,
                       num_frames, C);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
