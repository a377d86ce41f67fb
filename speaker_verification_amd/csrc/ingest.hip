// Audio ingest: interleaved 16-bit PCM at any rate -> mono samples at the model's rate, on device.
//
// Replaces the load step of /root/reference/utils.py:170-173
//     audio, sr = librosa.load(filename, sr=16000, mono=True)
// (down-mix = mean over channels, scale int16 / 32768 to [-1, 1), resample to 16 kHz).  librosa
// is absent here and its resampler unpinned, so the resampler is the published polyphase
// windowed-sinc scheme of scipy.signal.resample_poly (zero-padded edges): with
//     h = up * firwin(2 * half + 1, 1 / max(up, down), window = ("kaiser", 5.0)),  half = 10 max(up, down)
//     y[m] = sum_j x[j] * h[m * down + half - j * up]
// The taps come from the host (float64 -> float32, speaker_verification_amd/ingest.py), like the
// filterbank does.  One workgroup produces 256 consecutive output samples of one clip: the input
// span they touch is down-mixed once into LDS, the taps sit in LDS too, every lane then runs its
// dot product out of LDS.  HBM-bound: 2 n_ch bytes read per input frame, 4 (or 2) written per
// output sample.
#include "svk_internal.h"

namespace {

constexpr int OUT_PER_WG = 256;

template <typename OutT>
__global__ __launch_bounds__(256) void resample_kernel(const int16_t* __restrict__ pcm, int n_ch, int64_t in_stride,
                                                       const int32_t* __restrict__ in_len, int clip_in,
                                                       const float* __restrict__ taps, int n_taps, int up, int down,
                                                       OutT* __restrict__ out, int64_t out_stride, int clip_out,
                                                       int32_t* __restrict__ out_len, int span_cap) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h = smem;            // [n_taps]
  float* xs = smem + n_taps;  // [span_cap] mono samples, already / 32768
  const int utt = blockIdx.y;
  const int n_in = in_len ? in_len[utt] : clip_in;
  // resample_poly: ceil(n_in * up / down) output samples
  const int64_t n_out64 = ((int64_t)n_in * up + down - 1) / down;
  const int n_out = n_out64 < clip_out ? (int)n_out64 : clip_out;
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_len) out_len[utt] = n_out;
  const int m0 = blockIdx.x * OUT_PER_WG;
  OutT* o = out + (int64_t)utt * out_stride;
  if (m0 >= n_out) {  // past the clip: defined (zero) output
    for (int m = m0 + threadIdx.x; m < m0 + OUT_PER_WG && m < clip_out; m += blockDim.x) o[m] = (OutT)0;
    return;
  }
  const int half = (n_taps - 1) / 2;
  for (int i = threadIdx.x; i < n_taps; i += blockDim.x) h[i] = taps[i];
  // input frames touched by outputs m0 .. m0 + 255:  ceil((m down - half) / up) .. floor((m down + half) / up)
  const int64_t t_lo = (int64_t)m0 * down - half, t_hi = (int64_t)(m0 + OUT_PER_WG - 1) * down + half;
  const int64_t j_lo = t_lo >= 0 ? (t_lo + up - 1) / up : -((-t_lo) / up);
  const int64_t j_hi = t_hi / up;
  const int span = (int)(j_hi - j_lo + 1);  // <= span_cap by construction of the launch
  const int16_t* x = pcm + (int64_t)utt * in_stride * n_ch;
  const float scale = 1.0f / (32768.0f * (float)n_ch);
  for (int i = threadIdx.x; i < span; i += blockDim.x) {
    const int64_t j = j_lo + i;
    float v = 0.f;
    if (j >= 0 && j < n_in) {
      int acc = 0;
      for (int c = 0; c < n_ch; ++c) acc += x[j * n_ch + c];
      v = (float)acc * scale;
    }
    xs[i] = v;
  }
  __syncthreads();
  const int m = m0 + threadIdx.x;
  if (m >= clip_out) return;
  float y = 0.f;
  if (m < n_out) {
    const int64_t t = (int64_t)m * down;
    const int64_t a = t - half, b = t + half;
    const int64_t ja = a >= 0 ? (a + up - 1) / up : -((-a) / up);
    const int64_t jb = b / up;
    int k = (int)(t + half - ja * up);  // tap index of the first frame; steps down by `up`
    for (int64_t j = ja; j <= jb; ++j, k -= up) y = fmaf(xs[j - j_lo], h[k], y);
  }
  if constexpr (sizeof(OutT) == 2) {
    float r = rintf(y * 32768.0f);  // back to the int16 grid, round half to even, saturating
    r = r > 32767.f ? 32767.f : (r < -32768.f ? -32768.f : r);
    o[m] = (OutT)r;
  } else {
    o[m] = y;
  }
}

// Integer decimation (up == 1: 48 kHz or 32 kHz -> 16 kHz, the common cases) with the default
// design (n_taps = 20 DOWN + 1).  Every output then uses the SAME taps, so they are wave-uniform
// (scalar loads, no LDS traffic), and a lane that computes R = 3 consecutive outputs reads its
// 60 DOWN + 1 + 2 DOWN input samples from LDS once into registers instead of once per tap: LDS
// reads per multiply-add drop from 2 to 0.37 and the kernel moves from LDS-bound to HBM-bound.
// (R DOWN = 9 floats between lanes for DOWN = 3: conflict-free ds_read_b32.)
template <int DOWN, typename OutT>
__global__ __launch_bounds__(256) void decimate_kernel(const int16_t* __restrict__ pcm, int n_ch, int64_t in_stride,
                                                       const int32_t* __restrict__ in_len, int clip_in,
                                                       const float* __restrict__ taps, OutT* __restrict__ out,
                                                       int64_t out_stride, int clip_out, int32_t* __restrict__ out_len) {
  constexpr int HALF = 10 * DOWN, NT = 2 * HALF + 1, R = 3, OUT_WG = 256 * R;
  constexpr int SPAN = (OUT_WG - 1) * DOWN + NT + 8;  // + 8: the staged span starts on a 16-byte boundary
  __shared__ __attribute__((aligned(16))) float xs[SPAN];
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  const int utt = blockIdx.y;
  const int n_in = in_len ? in_len[utt] : clip_in;
  const int64_t n_out64 = ((int64_t)n_in + DOWN - 1) / DOWN;
  const int n_out = n_out64 < clip_out ? (int)n_out64 : clip_out;
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_len) out_len[utt] = n_out;
  const int m0 = blockIdx.x * OUT_WG;
  OutT* o = out + (int64_t)utt * out_stride;
  if (m0 >= n_out) {
    for (int m = m0 + threadIdx.x; m < m0 + OUT_WG && m < clip_out; m += blockDim.x) o[m] = (OutT)0;
    return;
  }
  const int16_t* x = pcm + (int64_t)utt * in_stride * n_ch;
  const int64_t j_lo = (int64_t)m0 * DOWN - HALF;
  const int64_t j_al = j_lo & ~(int64_t)7;  // floor to a multiple of 8 frames (also for negative j_lo)
  const int lead = (int)(j_lo - j_al);      // xs[lead + i] = x[j_lo + i]
  const float scale = 1.0f / (32768.0f * (float)n_ch);
  const bool vec = n_ch == 1 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  for (int v = threadIdx.x; 8 * v < SPAN; v += blockDim.x) {
    const int64_t j = j_al + 8 * v;
    float f[8];
    if (vec && j >= 0 && j + 8 <= n_in) {
      const i16x8 raw = *reinterpret_cast<const i16x8*>(x + j);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (float)raw[e] * scale;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t je = j + e;
        int acc = 0;
        if (je >= 0 && je < n_in)
          for (int c = 0; c < n_ch; ++c) acc += x[je * n_ch + c];
        f[e] = (float)acc * scale;
      }
    }
    if (8 * v + 8 <= SPAN) {  // two 16-byte stores (eight scalar ones at a stride of 8 floats between lanes: 16-way bank conflicts)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f32x4*>(xs + 8 * v) = (f32x4){f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(xs + 8 * v + 4) = (f32x4){f[4], f[5], f[6], f[7]};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (8 * v + e < SPAN) xs[8 * v + e] = f[e];
    }
  }
  __syncthreads();
  const int m = m0 + R * threadIdx.x;
  if (m >= clip_out) return;
  float xw[NT + (R - 1) * DOWN];
  const float* xp = xs + lead + R * DOWN * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NT + (R - 1) * DOWN; ++i) xw[i] = xp[i];
  float y[R];
#pragma unroll
  for (int r = 0; r < R; ++r) y[r] = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float hn = taps[NT - 1 - n];  // wave-uniform: a scalar load
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = fmaf(xw[r * DOWN + n], hn, y[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (m + r >= clip_out) break;
    float v = m + r < n_out ? y[r] : 0.f;
    if constexpr (sizeof(OutT) == 2) {
      v = rintf(v * 32768.0f);
      v = v > 32767.f ? 32767.f : (v < -32768.f ? -32768.f : v);
      o[m + r] = (OutT)v;
    } else {
      o[m + r] = v;
    }
  }
}

template <int DOWN, typename OutT>
void launch_decimate(svk_ctx* ctx, const int16_t* d_pcm, int n_ch, int64_t in_stride, const int32_t* d_in_len,
                     int clip_in, int n_utt, const float* d_taps, void* d_out, int64_t out_stride, int clip_out,
                     int32_t* d_out_len) {
  const dim3 grid((unsigned)((clip_out + 256 * 3 - 1) / (256 * 3)), (unsigned)n_utt);
  hipLaunchKernelGGL((decimate_kernel<DOWN, OutT>), grid, dim3(256), 0, ctx->stream, d_pcm, n_ch, in_stride, d_in_len,
                     clip_in, d_taps, static_cast<OutT*>(d_out), out_stride, clip_out, d_out_len);
}

}  // namespace

extern "C" {

int svk_ingest_resample(svk_ctx* ctx, const int16_t* d_pcm, int32_t n_ch, int64_t in_stride, const int32_t* d_in_len,
                        int32_t clip_in, int32_t n_utt, const float* d_taps, int32_t n_taps, int32_t up, int32_t down,
                        void* d_out, int32_t out_dtype, int64_t out_stride, int32_t clip_out, int32_t* d_out_len) {
  SVK_REQUIRE(ctx, ctx != nullptr, "ctx");
  if (n_utt == 0 || clip_out == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_pcm && d_taps && d_out, "null buffer");
  SVK_REQUIRE(ctx, n_utt > 0 && n_ch >= 1 && n_ch <= 8 && clip_in >= 0 && clip_out > 0, "shape");
  SVK_REQUIRE(ctx, in_stride >= clip_in && out_stride >= clip_out, "strides shorter than the clips");
  SVK_REQUIRE(ctx, up >= 1 && down >= 1 && n_taps >= 1 && (n_taps & 1) == 1, "up, down >= 1 and an odd tap count");
  SVK_REQUIRE(ctx, out_dtype == SVK_PCM_I16 || out_dtype == SVK_PCM_F32, "out_dtype");
  if (up == 1 && (down == 2 || down == 3) && n_taps == 20 * down + 1) {  // integer decimation, default design
    const bool i16 = out_dtype == SVK_PCM_I16;
    if (down == 2) {
      if (i16) launch_decimate<2, int16_t>(ctx, d_pcm, n_ch, in_stride, d_in_len, clip_in, n_utt, d_taps, d_out, out_stride, clip_out, d_out_len);
      else launch_decimate<2, float>(ctx, d_pcm, n_ch, in_stride, d_in_len, clip_in, n_utt, d_taps, d_out, out_stride, clip_out, d_out_len);
    } else {
      if (i16) launch_decimate<3, int16_t>(ctx, d_pcm, n_ch, in_stride, d_in_len, clip_in, n_utt, d_taps, d_out, out_stride, clip_out, d_out_len);
      else launch_decimate<3, float>(ctx, d_pcm, n_ch, in_stride, d_in_len, clip_in, n_utt, d_taps, d_out, out_stride, clip_out, d_out_len);
    }
    SVK_LAUNCH_CHECK(ctx);
    return SVK_OK;
  }
  const int half = (n_taps - 1) / 2;
  // frames one workgroup can touch: ((256 - 1) down + 2 half) / up + 2
  const int64_t span_cap = ((int64_t)(OUT_PER_WG - 1) * down + 2LL * half) / up + 2;
  const size_t lds = sizeof(float) * ((size_t)n_taps + (size_t)span_cap);
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "resampling %d/%d needs %zu bytes of LDS per workgroup (have %d)", up,
                    down, lds, ctx->lds_per_cu);
  const dim3 grid((unsigned)((clip_out + OUT_PER_WG - 1) / OUT_PER_WG), (unsigned)n_utt);
  if (out_dtype == SVK_PCM_I16) {
    auto kern = resample_kernel<int16_t>;
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, d_pcm, n_ch, in_stride, d_in_len, clip_in, d_taps,
                       n_taps, up, down, static_cast<int16_t*>(d_out), out_stride, clip_out, d_out_len, (int)span_cap);
  } else {
    auto kern = resample_kernel<float>;
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, d_pcm, n_ch, in_stride, d_in_len, clip_in, d_taps,
                       n_taps, up, down, static_cast<float*>(d_out), out_stride, clip_out, d_out_len, (int)span_cap);
  }
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

}  // extern "C"
