// Stage-level entry points: one kernel per SpeechPy function, so that every function of
// the reference's processing module can be routed to the GPU on its own.
//   svk_preemphasis  <- processing.py:45-58      svk_stack_frames <- processing.py:61-139
//   svk_spectrum     <- processing.py:142-174    svk_cmvn         <- processing.py:239-271
//   svk_cube_gather  <- /root/reference/utils.py:351-379
// All of these are HBM-streaming kernels: 16-byte accesses where alignment allows,
// grid capped at a few workgroups per CU with a grid-stride loop.
#include <algorithm>

#include "fft_wave.h"
#include <cstdlib>

#include "svk_internal.h"

using namespace svk_fft;
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

inline unsigned capped_grid(const svk_ctx* ctx, int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  int64_t cap = (int64_t)ctx->num_cu * 8;
  return (unsigned)std::max<int64_t>(1, std::min(blocks, cap));
}

// ---- pre-emphasis ---------------------------------------------------------------
template <typename PcmT>
__global__ __launch_bounds__(256) void preemph_kernel(const PcmT* __restrict__ x, int64_t n, int64_t shift_mod,
                                                      float cof, float* __restrict__ y) {
  // shift_mod = shift mod n in [0, n): y[i] = x[i] - cof * x[(i - shift) mod n]
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = i - shift_mod;
    if (j < 0) j += n;
    y[i] = (float)x[i] - cof * (float)x[j];
  }
}

// ---- framing ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void stack_frames_kernel(const float* __restrict__ sig, int64_t n, int flen,
                                                           int stride, int64_t total, const float* __restrict__ win,
                                                           float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / flen;
    const int j = (int)(i - t * flen);
    const int64_t s = t * stride + j;
    float v = s < n ? sig[s] : 0.f;  // zero padding of processing.py:107-109
    if (win) v *= win[j];
    out[i] = v;
  }
}

// ---- spectrum: wave FFT for nfft 512 / 1024 ------------------------------------------
// Four independent waves per workgroup, each with its own scratch (they never wait for each other);
// nfft = 512 handles two frames per FFT, nfft = 1024 one.
constexpr int SPEC_WAVES = 4;
template <bool SPLIT1024>
__global__ __launch_bounds__(64 * SPEC_WAVES) void spectrum_fft_kernel(const float* __restrict__ frames, int nframes,
                                                                       int flen, int power,
                                                                       const cplx* __restrict__ tw,
                                                                       float* __restrict__ out) {
  __shared__ cplx scr_all[SPEC_WAVES][SCR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  cplx* scr = scr_all[wave];
  const int nfft = SPLIT1024 ? 1024 : 512;
  const int nbins = nfft / 2 + 1;
  const int feff = flen < nfft ? flen : nfft;
  cplx t1[8], t2[8], t3[5];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    t1[r] = tw[r * 64 + lane];
    t2[r] = tw[512 + r * 64 + lane];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) t3[q] = tw[1024 + q * 64 + lane];
  const int mirror = (64 - lane) & 63;
  const bool lane0 = lane == 0;
  constexpr int PER = SPLIT1024 ? 1 : 2;
  const float scale = power ? (SPLIT1024 ? 1.f / 1024.f : 1.f / 2048.f) : (SPLIT1024 ? 1.f : 0.25f);
  for (int fa = (blockIdx.x * SPEC_WAVES + wave) * PER; fa < nframes; fa += gridDim.x * SPEC_WAVES * PER) {
    const float* sa = frames + (int64_t)fa * flen;
    const bool hasb = !SPLIT1024 && fa + 1 < nframes;
    const float* sb = frames + (int64_t)(hasb ? fa + 1 : fa) * flen;
    cplx v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (SPLIT1024) {
        const int i0 = 2 * (lane + 64 * a);
        v[a].x = i0 < feff ? sa[i0] : 0.f;
        v[a].y = i0 + 1 < feff ? sa[i0 + 1] : 0.f;
      } else {
        const int nn = lane + 64 * a;
        v[a].x = nn < feff ? sa[nn] : 0.f;
        v[a].y = (nn < feff && hasb) ? sb[nn] : 0.f;
      }
    }
    fft512_wave(v, scr, lane, t1, t2);
    cplx s7 = shfl2(v[7], mirror), s6 = shfl2(v[6], mirror), s5 = shfl2(v[5], mirror), s4 = shfl2(v[4], mirror);
    cplx zm[5] = {lane0 ? v[0] : s7, lane0 ? s7 : s6, lane0 ? s6 : s5, lane0 ? s5 : s4, v[4]};
    float* oa = out + (int64_t)fa * nbins;
    float* ob = oa + nbins;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 4 && !lane0) continue;
      const cplx zk = v[j], zn = zm[j];
      const int k = j == 4 ? 256 : lane + 64 * j;
      if (SPLIT1024) {
        const cplx E = mk(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const cplx O = mk(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const cplx Tw = cmul(t3[j], O);
        const cplx xp = E + Tw, xm = E - Tw;
        float pk = (xp.x * xp.x + xp.y * xp.y) * scale, pm = (xm.x * xm.x + xm.y * xm.y) * scale;
        if (!power) {
          pk = sqrtf(pk);
          pm = sqrtf(pm);
        }
        oa[k] = pk;
        if (j != 4) oa[512 - k] = pm;
      } else {
        const float ar = zk.x + zn.x, ai = zk.y - zn.y, br = zk.y + zn.y, bi = zk.x - zn.x;
        float pa = (ar * ar + ai * ai) * scale, pb = (br * br + bi * bi) * scale;
        if (!power) {
          pa = sqrtf(pa);
          pb = sqrtf(pb);
        }
        oa[k] = pa;
        if (hasb) ob[k] = pb;
      }
    }
  }
}

// ---- spectrum: any other power of two (4 .. 8192) -- Stockham radix-2 FFT in LDS -----------------------
// The real frame of nfft samples is packed as M = nfft/2 complex numbers (even, odd samples), transformed by
// log2(M) autosort passes between two LDS buffers and untangled (X[k] = E[k] + W_nfft^k O[k]).  A frame is
// worked on by TPF = min(256, M/2) threads, 256 / TPF frames per workgroup; `tw` holds W_nfft^i, i < nfft/2
// (f64-computed, rounded once): pass twiddles W_M^j are its even entries.
__global__ __launch_bounds__(256) void spectrum_pow2_kernel(const float* __restrict__ frames, int nframes, int flen,
                                                            int nfft, int power, const cplx* __restrict__ tw_g,
                                                            float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_fft[];
  const int M = nfft >> 1, nbins = M + 1;
  const int feff = flen < nfft ? flen : nfft;
  const int tpf = M / 2 < 256 ? (M / 2 < 1 ? 1 : M / 2) : 256;  // threads per frame
  const int fpb = 256 / tpf;                                    // frames per workgroup
  cplx* tw = reinterpret_cast<cplx*>(smem_fft);                 // [M]  W_nfft^i
  cplx* buf = tw + M;                                           // [fpb][2][M]
  for (int i = threadIdx.x; i < M; i += 256) tw[i] = tw_g[i];
  const int sub = threadIdx.x / tpf, t = threadIdx.x - sub * tpf;
  cplx* a = buf + (size_t)sub * 2 * M;
  cplx* b = a + M;
  const float scale = power ? 1.0f / (float)nfft : 1.0f;
  for (int64_t f0 = (int64_t)blockIdx.x * fpb; f0 < nframes; f0 += (int64_t)gridDim.x * fpb) {
    const int64_t f = f0 + sub;
    const bool live = f < nframes;
    const float* src = frames + (live ? f : 0) * flen;
    __syncthreads();  // previous frame's readers are done (and the table is in place)
    for (int j = t; j < M; j += tpf) {
      const int i0 = 2 * j;
      a[j] = mk(live && i0 < feff ? src[i0] : 0.f, live && i0 + 1 < feff ? src[i0 + 1] : 0.f);
    }
    cplx* x = a;
    cplx* y = b;
    for (int ns = 1; ns < M; ns <<= 1) {
      __syncthreads();
      const int tstep = M / ns;  // W_{2 ns}^k = W_nfft^(k * nfft / (2 ns)) = tw[k * M / ns]
      for (int j = t; j < M / 2; j += tpf) {
        const int k = j & (ns - 1);
        const cplx w = tw[k * tstep];
        const cplx u = x[j], v0 = x[j + M / 2];
        const cplx v = mk(v0.x * w.x - v0.y * w.y, v0.x * w.y + v0.y * w.x);
        const int j0 = ((j - k) << 1) + k;
        y[j0] = u + v;
        y[j0 + ns] = u - v;
      }
      cplx* tmp = x;
      x = y;
      y = tmp;
    }
    __syncthreads();
    // untangle: Z = FFT_M(even + i odd);  E[k] = (Z[k] + conj Z[M-k]) / 2,  O[k] = (Z[k] - conj Z[M-k]) / (2i)
    if (live) {
      float* o = out + f * nbins;
      for (int k = t; k <= M; k += tpf) {
        const cplx zk = x[k == M ? 0 : k], zn = x[k == 0 ? 0 : M - k];
        const float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);
        const float orr = 0.5f * (zk.y + zn.y), oi = -0.5f * (zk.x - zn.x);
        const cplx w = k == M ? mk(-1.f, 0.f) : tw[k];
        const float xr = er + orr * w.x - oi * w.y, xi = ei + orr * w.y + oi * w.x;
        const float p = (xr * xr + xi * xi) * scale;
        o[k] = power ? p : sqrtf(p);
      }
    }
  }
}

// ---- spectrum: every other length -- direct DFT, O(n^2), phases from a table ----------------------------
// `tw` holds W_nfft^i for i < nfft (f64-computed, rounded once to f32); a workgroup keeps it and FR frames
// in LDS; thread -> output bin k walks the phase index k n mod nfft by addition (no division, no sincos
// in the loop) with float64 accumulators.
constexpr int DFT_FR = 4;
__global__ __launch_bounds__(256) void spectrum_dft_kernel(const float* __restrict__ frames, int nframes, int flen,
                                                           int nfft, int power, const cplx* __restrict__ tw_g,
                                                           float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_dft[];
  const int nbins = nfft / 2 + 1;
  const int feff = flen < nfft ? flen : nfft;
  cplx* tw = reinterpret_cast<cplx*>(smem_dft);                 // [nfft]
  float* x = reinterpret_cast<float*>(tw + nfft);              // [DFT_FR][feff]
  for (int i = threadIdx.x; i < nfft; i += 256) tw[i] = tw_g[i];
  for (int64_t f0 = (int64_t)blockIdx.x * DFT_FR; f0 < nframes; f0 += (int64_t)gridDim.x * DFT_FR) {
    __syncthreads();
    for (int i = threadIdx.x; i < DFT_FR * feff; i += 256) {
      const int q = i / feff, n = i - q * feff;
      x[i] = f0 + q < nframes ? frames[(f0 + q) * flen + n] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256) {
      double re[DFT_FR] = {0.0}, im[DFT_FR] = {0.0};
      int ph = 0;
      for (int n = 0; n < feff; ++n) {
        const cplx w = tw[ph];
        ph += k;
        ph = ph >= nfft ? ph - nfft : ph;
#pragma unroll
        for (int q = 0; q < DFT_FR; ++q) {
          const double v = (double)x[q * feff + n];
          re[q] += v * (double)w.x;
          im[q] += v * (double)w.y;
        }
      }
#pragma unroll
      for (int q = 0; q < DFT_FR; ++q)
        if (f0 + q < nframes) {
          const double mag2 = re[q] * re[q] + im[q] * im[q];
          out[(f0 + q) * nbins + k] = (float)(power ? mag2 / (double)nfft : sqrt(mag2));
        }
    }
  }
}

// W_n^i = exp(-2 pi i / n), i < count, computed on the device in float64 (one launch per call: the table
// depends on nfft and lives in the handle's workspace)
__global__ __launch_bounds__(256) void twiddle_table_kernel(cplx* __restrict__ tw, int n, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    double sn, cs;
    sincospi(-2.0 * (double)i / (double)n, &sn, &cs);
    tw[i] = mk((float)cs, (float)sn);
  }
}

// ---- CMVN: one workgroup per clip ------------------------------------------------------
// Threads form a [R rows][cb cols] grid over a column block; column sums are kept in float64.
// APPLY = false: the clip's statistics only, stats[utt][0][c] = mean, stats[utt][1][c] = 1 / (std + 2^-30) (svk_cmvn_stats:
// the normalisation is then applied where the rows are next read, svk_cube_gather_cmvn).
template <bool APPLY>
__global__ __launch_bounds__(256) void cmvn_kernel(float* __restrict__ feat, int max_frames, int ncols,
                                                   const int32_t* __restrict__ n_frames, int variance,
                                                   double* __restrict__ stats) {
  __shared__ double red[256];
  const int utt = blockIdx.x;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  if (T <= 0) return;
  float* base = feat + (int64_t)utt * max_frames * ncols;
  for (int c0 = 0; c0 < ncols; c0 += 256) {
    const int cb = min(256, ncols - c0);  // columns in this block
    const int R = 256 / cb;               // row groups
    const int tc = threadIdx.x % cb, tr = threadIdx.x / cb;
    const bool active = tr < R;
    // ONE pass: column sums and sums of squares in float64 (the long-clip path's formula: var = E[x^2] - mean^2; for log-mel /
    // cepstral magnitudes the cancellation costs ~1e-14 relative in float64, far below the float32 the result is stored in)
    double s1 = 0.0, s2 = 0.0;
    if (active) {
      // four rows' loads in flight per thread (one load per trip followed by its own use left the pass latency-bound:
      // 1.3 TB/s on 7 s clips); the sums stay in row order
      const float* col = base + c0 + tc;
      int t = tr;
      for (; t + 3 * R < T; t += 4 * R) {
        const float v0 = col[(int64_t)t * ncols], v1 = col[(int64_t)(t + R) * ncols], v2 = col[(int64_t)(t + 2 * R) * ncols],
                    v3 = col[(int64_t)(t + 3 * R) * ncols];
        s1 += (double)v0;
        s2 += (double)v0 * (double)v0;
        s1 += (double)v1;
        s2 += (double)v1 * (double)v1;
        s1 += (double)v2;
        s2 += (double)v2 * (double)v2;
        s1 += (double)v3;
        s2 += (double)v3 * (double)v3;
      }
      for (; t < T; t += R) {
        const double v = (double)col[(int64_t)t * ncols];
        s1 += v;
        s2 += v * v;
      }
    }
    red[threadIdx.x] = active ? s1 : 0.0;
    __syncthreads();
    double mean = 0.0;
    if (active) {
      for (int r = 0; r < R; ++r) mean += red[r * cb + tc];
      mean /= (double)T;
    }
    __syncthreads();
    double inv = 1.0;
    if (variance) {
      red[threadIdx.x] = active ? s2 : 0.0;
      __syncthreads();
      if (active) {
        double q = 0.0;
        for (int r = 0; r < R; ++r) q += red[r * cb + tc];
        double var = q / (double)T - mean * mean;
        var = var > 0.0 ? var : 0.0;
        inv = 1.0 / (sqrt(var) + 9.313225746154785e-10);  // + 2^-30, processing.py:250,266
      }
      __syncthreads();
    }
    if (APPLY) {
      if (active) {
        float* col = base + c0 + tc;
        int t = tr;
        for (; t + 3 * R < T; t += 4 * R) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = col[(int64_t)(t + k * R) * ncols];
#pragma unroll
          for (int k = 0; k < 4; ++k) col[(int64_t)(t + k * R) * ncols] = (float)(((double)v[k] - mean) * inv);
        }
        for (; t < T; t += R) col[(int64_t)t * ncols] = (float)(((double)col[(int64_t)t * ncols] - mean) * inv);
      }
    } else if (active && tr == 0) {
      stats[((int64_t)utt * 2) * ncols + c0 + tc] = mean;
      stats[((int64_t)utt * 2 + 1) * ncols + c0 + tc] = inv;
    }
    __syncthreads();
  }
}

// ---- CMVN for LONG clips (VoxCeleb utterances run to 145 s = 14 500 frames): with one workgroup per clip a batch of a
// few long clips leaves the chip to a handful of workgroups that each walk megabytes three times (measured: 1.07 ms for
// 14 clips of 31 .. 145 s).  Three small kernels cut a clip into chunks of CMVN_CHUNK frames instead: per-chunk column
// sums and sums of squares in float64, a per-clip reduction over the chunks IN ORDER (bitwise repeatable; no atomics)
// to mean and 1 / (std + 2^-30), and the normalisation.  var = E[x^2] - mean^2 in float64: for features of magnitude
// <= 10^2 and f32 inputs the cancellation costs < 1e-12 relative -- far below the float32 the result is stored in. ----
constexpr int CMVN_CHUNK = 256;

__global__ __launch_bounds__(256) void cmvn_partial_kernel(const float* __restrict__ feat, int max_frames, int ncols,
                                                           const int32_t* __restrict__ n_frames, int n_chunks,
                                                           double* __restrict__ part /* [utt][chunk][2][ncols] */) {
  __shared__ double red[2][256];
  const int utt = blockIdx.y, chunk = blockIdx.x;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  const int t0 = chunk * CMVN_CHUNK, t1 = min(T, t0 + CMVN_CHUNK);
  const float* base = feat + (int64_t)utt * max_frames * ncols;
  double* out = part + ((int64_t)utt * n_chunks + chunk) * 2 * ncols;
  for (int c0 = 0; c0 < ncols; c0 += 256) {
    const int cb = min(256, ncols - c0), R = 256 / cb;
    const int tc = threadIdx.x % cb, tr = threadIdx.x / cb;
    double s = 0.0, q = 0.0;
    if (tr < R)
      for (int t = t0 + tr; t < t1; t += R) {
        const double v = (double)base[(int64_t)t * ncols + c0 + tc];
        s += v;
        q += v * v;
      }
    red[0][threadIdx.x] = tr < R ? s : 0.0;
    red[1][threadIdx.x] = tr < R ? q : 0.0;
    __syncthreads();
    if (tr == 0) {
      double ss = 0.0, qq = 0.0;
      for (int r = 0; r < R; ++r) {
        ss += red[0][r * cb + tc];
        qq += red[1][r * cb + tc];
      }
      out[c0 + tc] = ss;
      out[ncols + c0 + tc] = qq;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void cmvn_stats_kernel(const double* __restrict__ part, int max_frames, int ncols,
                                                         const int32_t* __restrict__ n_frames, int n_chunks, int variance,
                                                         double* __restrict__ stats /* [utt][2][ncols]: mean, inv */) {
  const int utt = blockIdx.x;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  if (T <= 0) return;
  const int used = (T + CMVN_CHUNK - 1) / CMVN_CHUNK;
  for (int c = threadIdx.x; c < ncols; c += 256) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < used; ++k) {        // fixed order: repeatable
      const double* pk = part + ((int64_t)utt * n_chunks + k) * 2 * ncols;
      s += pk[c];
      q += pk[ncols + c];
    }
    const double mean = s / (double)T;
    double inv = 1.0;
    if (variance) {
      double var = q / (double)T - mean * mean;
      var = var > 0.0 ? var : 0.0;
      inv = 1.0 / (sqrt(var) + 9.313225746154785e-10);   // + 2^-30, processing.py:250,266
    }
    stats[((int64_t)utt * 2) * ncols + c] = mean;
    stats[((int64_t)utt * 2 + 1) * ncols + c] = inv;
  }
}

__global__ __launch_bounds__(256) void cmvn_apply_kernel(float* __restrict__ feat, int max_frames, int ncols,
                                                         const int32_t* __restrict__ n_frames,
                                                         const double* __restrict__ stats) {
  const int utt = blockIdx.y, chunk = blockIdx.x;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  const int t0 = chunk * CMVN_CHUNK, t1 = min(T, t0 + CMVN_CHUNK);
  if (t0 >= t1) return;
  float* base = feat + ((int64_t)utt * max_frames + t0) * ncols;
  const double* st = stats + (int64_t)utt * 2 * ncols;
  const int n = (t1 - t0) * ncols;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int c = e % ncols;
    base[e] = (float)(((double)base[e] - st[c]) * st[ncols + c]);
  }
}

// ---- general mel / log / DCT stage (any fft length, any bank) ------------------------------------------------
// mel[T][nf] = power[T][nbins] x bank^T: a workgroup owns MEL_FR frames and walks the filters in blocks of 64
// and the bins in chunks of 64, both operands staged in LDS with coalesced loads (the bank row stride + 1
// makes the per-filter reads conflict-free; the power value is a wave-wide broadcast).  Thread = (filter of
// the block, group of 4 frames).  The frame energy (sum over ALL bins, feature.py:202) is taken from the
// staged power chunks of the first filter block.  (log) mel energies of the tile stay in LDS; the DCT-II
// (ortho) rows come from a table built once per call (dct_table_kernel).
constexpr int MEL_FR = 16, MEL_CH = 64;
__global__ __launch_bounds__(256) void mel_features_kernel(const float* __restrict__ power, int nframes, int nbins,
                                                           const float* __restrict__ bank, int nf, int out_kind,
                                                           int ncep, int dc_elim, const float* __restrict__ dct,
                                                           float* __restrict__ feat, float* __restrict__ energy) {
  extern __shared__ __attribute__((aligned(16))) char smem_mel[];
  float* ptile = reinterpret_cast<float*>(smem_mel);      // [MEL_FR][MEL_CH]
  float* btile = ptile + MEL_FR * MEL_CH;                  // [64][MEL_CH + 1]
  float* etot = btile + 64 * (MEL_CH + 1);                 // [MEL_FR]
  float* lmel = etot + MEL_FR;                             // [MEL_FR][nf]
  const float EPS = 2.220446049250313e-16f;
  const int t = threadIdx.x, fi = t & 63, fg = t >> 6;     // filter inside the block, frame group (wave-uniform)
  const int64_t f0 = (int64_t)blockIdx.x * MEL_FR;
  const int nfr = (int)(nframes - f0 < MEL_FR ? nframes - f0 : MEL_FR);
  float esum = 0.f;                                        // thread (frame t / 16, part t % 16): energy partial
  for (int fb = 0; fb < nf; fb += 64) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nbins; k0 += MEL_CH) {
      __syncthreads();
      for (int i = t; i < MEL_FR * MEL_CH; i += 256) {
        const int r = i / MEL_CH, k = i - r * MEL_CH;
        ptile[i] = (r < nfr && k0 + k < nbins) ? power[(f0 + r) * nbins + k0 + k] : 0.f;
      }
      for (int i = t; i < 64 * MEL_CH; i += 256) {
        const int r = i / MEL_CH, k = i - r * MEL_CH;
        btile[r * (MEL_CH + 1) + k] = (fb + r < nf && k0 + k < nbins) ? bank[(int64_t)(fb + r) * nbins + k0 + k] : 0.f;
      }
      __syncthreads();
      if (fb == 0) {
        const float* pr = ptile + (t >> 4) * MEL_CH + (t & 15) * 4;
        esum += (pr[0] + pr[1]) + (pr[2] + pr[3]);
      }
      const float* br = btile + fi * (MEL_CH + 1);
      const float* pr = ptile + fg * 4 * MEL_CH;
#pragma unroll 8
      for (int k = 0; k < MEL_CH; ++k) {
        const float w = br[k];
        acc[0] = fmaf(pr[k], w, acc[0]);
        acc[1] = fmaf(pr[MEL_CH + k], w, acc[1]);
        acc[2] = fmaf(pr[2 * MEL_CH + k], w, acc[2]);
        acc[3] = fmaf(pr[3 * MEL_CH + k], w, acc[3]);
      }
    }
    if (fb + fi < nf) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float m = acc[q] == 0.f ? EPS : acc[q];  // feature.py:217
        lmel[(fg * 4 + q) * nf + fb + fi] = out_kind == SVK_OUT_MFE ? m : logf(m);
      }
    }
  }
  // frame energies: 16 partials per frame sit in 16 consecutive lanes
  esum += __shfl_xor(esum, 8, 64);
  esum += __shfl_xor(esum, 4, 64);
  esum += __shfl_xor(esum, 2, 64);
  esum += __shfl_xor(esum, 1, 64);
  if ((t & 15) == 0) etot[t >> 4] = esum == 0.f ? EPS : esum;  // feature.py:205
  __syncthreads();
  if (energy && t < nfr) energy[f0 + t] = etot[t];
  const int cols = out_kind == SVK_OUT_MFCC ? ncep : nf;
  for (int i = t; i < nfr * cols; i += 256) {
    const int r = i / cols, cc = i - r * cols;
    float v;
    if (out_kind != SVK_OUT_MFCC) {
      v = lmel[r * nf + cc];
    } else if (cc == 0 && dc_elim) {
      v = logf(etot[r]);                             // feature.py:151-152
    } else {
      const float* d = dct + (int64_t)cc * nf;
      const float* lm = lmel + r * nf;
      float a0 = 0.f, a1 = 0.f;
      int n = 0;
      for (; n + 1 < nf; n += 2) {
        a0 = fmaf(lm[n], d[n], a0);
        a1 = fmaf(lm[n + 1], d[n + 1], a1);
      }
      if (n < nf) a0 = fmaf(lm[n], d[n], a0);
      v = a0 + a1;
    }
    feat[(f0 + r) * cols + cc] = v;
  }
}

// The same stage with the mel product on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32 like the fused front
// end's mel tile): a workgroup owns 64 frames (one 16-frame M tile per wave), walks the filters in blocks of 64 (four
// N tiles per wave) and the bins in chunks of 64; both operands are staged in LDS with rows of 68 floats (the 16 rows
// of a tile then start in 16 different 16-byte bank slots) and read as ds_read_b128 fragments -- K permuted identically
// on both (lane (i, kk) holds bins 16 g + 4 kk .. + 3, MFMA step e uses element e).  Used for banks of up to 256 filters
// (the 64 x nf tile of log-mel energies must fit LDS); larger banks take the VALU kernel above.
constexpr int MM_FR = 64, MM_CH = 64, MM_LD = 68;
typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mel_features_mfma_kernel(const float* __restrict__ power, int nframes, int nbins,
                                                                const float* __restrict__ bank, int nf, int out_kind,
                                                                int ncep, int dc_elim, const float* __restrict__ dct,
                                                                float* __restrict__ feat, float* __restrict__ energy) {
  extern __shared__ __attribute__((aligned(16))) char smem_mm[];
  float* ptile = reinterpret_cast<float*>(smem_mm);        // [64 frames][68]
  float* btile = ptile + MM_FR * MM_LD;                    // [64 filters][68]
  float* etot = btile + 64 * MM_LD;                        // [64]
  float* lmel = etot + MM_FR;                              // [64][nf]
  const float EPS = 2.220446049250313e-16f;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, kk = lane >> 4;
  const int64_t f0 = (int64_t)blockIdx.x * MM_FR;
  const int nfr = (int)(nframes - f0 < MM_FR ? nframes - f0 : MM_FR);
  float esum = 0.f;                                        // thread (frame t / 4, quarter t % 4 of a chunk): energy partial
  // Order: bins outermost, so a power chunk is staged once for all (up to four) filter blocks; the accumulators of every
  // block stay in registers.  The NEXT step's global loads are issued before this step's MFMAs (coalesced, element e =
  // t + 256 j -> (row e / 64, bin e % 64); zero beyond the frames / filters / bins).
  const int nfb = (nf + 63) >> 6;                           // 1 .. 4 filter blocks
  mm_f32x4 acc[4][4];
#pragma unroll
  for (int fbi = 0; fbi < 4; ++fbi)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[fbi][nt] = (mm_f32x4){0.f, 0.f, 0.f, 0.f};
  float pv[16], bv[16];
  auto load_power = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = t + 256 * j, r = e >> 6, k = e & 63;
      pv[j] = (r < nfr && k0 + k < nbins) ? power[(f0 + r) * nbins + k0 + k] : 0.f;
    }
  };
  auto load_bank = [&](int k0, int fb) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = t + 256 * j, r = e >> 6, k = e & 63;
      bv[j] = (fb + r < nf && k0 + k < nbins) ? bank[(int64_t)(fb + r) * nbins + k0 + k] : 0.f;
    }
  };
  load_power(0);
  load_bank(0, 0);
  for (int k0 = 0; k0 < nbins; k0 += MM_CH) {
#pragma unroll
    for (int fbi = 0; fbi < 4; ++fbi) {
      if (fbi >= nfb) break;
      __syncthreads();                                     // the previous step's fragment reads are done
      if (fbi == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ptile[((t + 256 * j) >> 6) * MM_LD + (t & 63)] = pv[j];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) btile[((t + 256 * j) >> 6) * MM_LD + (t & 63)] = bv[j];
      __syncthreads();
      if (fbi + 1 < nfb) {
        load_bank(k0, 64 * (fbi + 1));
      } else if (k0 + MM_CH < nbins) {
        load_power(k0 + MM_CH);
        load_bank(k0 + MM_CH, 0);
      }
      if (fbi == 0) {   // frame energy = the sum over ALL bins (feature.py:202)
        const float* pr = ptile + (t >> 2) * MM_LD + (t & 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const mm_f32x4 v = *reinterpret_cast<const mm_f32x4*>(pr + 4 * q);
          esum += (v[0] + v[1]) + (v[2] + v[3]);
        }
      }
      const float* ar = ptile + (16 * wave + i) * MM_LD + 4 * kk;
      const float* br = btile + i * MM_LD + 4 * kk;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const mm_f32x4 a = *reinterpret_cast<const mm_f32x4*>(ar + 16 * g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const mm_f32x4 b = *reinterpret_cast<const mm_f32x4*>(br + 16 * nt * MM_LD + 16 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[fbi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[fbi][nt], 0, 0, 0);
        }
      }
    }
  }
  // rows 4 kk + r = frame 16 wave + 4 kk + r, column i = filter 64 fbi + 16 nt + i
#pragma unroll
  for (int fbi = 0; fbi < 4; ++fbi)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (64 * fbi + 16 * nt + i < nf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m = acc[fbi][nt][r] == 0.f ? EPS : acc[fbi][nt][r];  // feature.py:217
          lmel[(16 * wave + 4 * kk + r) * nf + 64 * fbi + 16 * nt + i] = out_kind == SVK_OUT_MFE ? m : logf(m);
        }
      }
  esum += __shfl_xor(esum, 2, 64);
  esum += __shfl_xor(esum, 1, 64);
  if ((t & 3) == 0) etot[t >> 2] = esum == 0.f ? EPS : esum;  // feature.py:205
  __syncthreads();
  if (energy && t < nfr) energy[f0 + t] = etot[t];
  const int cols = out_kind == SVK_OUT_MFCC ? ncep : nf;
  for (int idx = t; idx < nfr * cols; idx += 256) {
    const int r = idx / cols, cc = idx - r * cols;
    float v;
    if (out_kind != SVK_OUT_MFCC) {
      v = lmel[r * nf + cc];
    } else if (cc == 0 && dc_elim) {
      v = logf(etot[r]);                             // feature.py:151-152
    } else {
      const float* d = dct + (int64_t)cc * nf;
      const float* lm = lmel + r * nf;
      float a0 = 0.f, a1 = 0.f;
      int n = 0;
      for (; n + 1 < nf; n += 2) {
        a0 = fmaf(lm[n], d[n], a0);
        a1 = fmaf(lm[n + 1], d[n + 1], a1);
      }
      if (n < nf) a0 = fmaf(lm[n], d[n], a0);
      v = a0 + a1;
    }
    feat[(f0 + r) * cols + cc] = v;
  }
}

// scipy.fftpack.dct(type=2, norm='ortho'): D[k][n] = sqrt(2/N) cos(pi k (2n+1) / 2N), D[0][n] = sqrt(1/N)
__global__ __launch_bounds__(256) void dct_table_kernel(float* __restrict__ d, int ncep, int nf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncep * nf) {
    const int k = i / nf, n = i - k * nf;
    d[i] = (float)(cospi((double)k * (2.0 * n + 1.0) / (2.0 * nf)) * (k == 0 ? sqrt(1.0 / nf) : sqrt(2.0 / nf)));
  }
}

// ---- sliding-window CMVN (cmvnw, Q10) ------------------------------------------------------
// 'symmetric' padding = reflect INCLUDING the edge sample, repeated when the window is longer
// than the clip: index k maps to m = k mod 2T, then m < T ? m : 2T - 1 - m.
__device__ __forceinline__ int sym_index(int k, int T) {
  int m = k % (2 * T);
  if (m < 0) m += 2 * T;
  return m < T ? m : 2 * T - 1 - m;
}

// pass 0: out = x - mean(window);  pass 1: out = src / (std(window of src) + 2^-30), src = pass-0 result.
// Thread = (clip, segment of `seg` consecutive rows, column): the window sums of the segment's first row are
// taken directly (win loads), every further row costs one row entering and one leaving the window: win / seg + 2
// loads per output instead of win (301 -> 11 at seg = 32), float64 running sums.
__global__ __launch_bounds__(256) void cmvnw_kernel(const float* __restrict__ src, int max_frames, int ncols,
                                                    const int32_t* __restrict__ n_frames, int win, int pass, int seg,
                                                    float* __restrict__ dst) {
  const int utt = blockIdx.y;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  if (T <= 0) return;
  const int half = (win - 1) / 2;
  const float* base = src + (int64_t)utt * max_frames * ncols;
  float* out = dst + (int64_t)utt * max_frames * ncols;
  const int nseg = (T + seg - 1) / seg;
  const double inv_win = 1.0 / (double)win;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nseg * ncols; i += gridDim.x * blockDim.x) {
    const int sg = i / ncols, c = i - sg * ncols;  // consecutive threads -> consecutive columns of one row
    const int r0 = sg * seg, r1 = r0 + seg < T ? r0 + seg : T;
    double s = 0.0, q = 0.0;
    for (int k = r0 - half; k <= r0 + half; ++k) {
      const double v = (double)base[(int64_t)sym_index(k, T) * ncols + c];
      s += v;
      q += v * v;
    }
    for (int r = r0; r < r1; ++r) {
      if (r > r0) {
        const double vin = (double)base[(int64_t)sym_index(r + half, T) * ncols + c];
        const double vout = (double)base[(int64_t)sym_index(r - half - 1, T) * ncols + c];
        s += vin - vout;
        q += vin * vin - vout * vout;
      }
      const double mean = s * inv_win;
      const double x = (double)base[(int64_t)r * ncols + c];
      if (pass == 0) {
        out[(int64_t)r * ncols + c] = (float)(x - mean);
      } else {
        double var = q * inv_win - mean * mean;
        var = var > 0.0 ? var : 0.0;
        out[(int64_t)r * ncols + c] = (float)(x / (sqrt(var) + 9.313225746154785e-10));
      }
    }
  }
}

// The same on clips that fit LDS (the reference's own use: a few hundred frames of 13 .. 40 coefficients, window 301 --
// longer than the clip): one workgroup per (clip, group of columns) keeps the columns in LDS, builds float64 PREFIX
// sums along time by wave-wide scans, and every window sum is then two look-ups: the symmetric padding is the
// periodic extension x[0 .. T-1], x[T-1 .. 0] of period 2 T, whose prefix sums follow from the clip's own
// (P2[m] = m <= T ? P[m] : 2 P[T] - P[2 T - m]; F(n) = floor(n / 2T) * 2 P[T] + P2[n mod 2T]).  Both passes (mean, then
// the std of the centred rows) run in the one kernel: one read + one write of the array instead of two + two, and no
// serial walk.  LDS: 20 bytes per element (the column as f32, prefix sums of y and y * y as f64).
constexpr int CW_CAP = 3900;   // elements per workgroup: 78 KB of LDS, two workgroups per CU
__device__ __forceinline__ double cw_prefix_at(const double* P, int T, int n) {   // F(n), n any integer
  const int two = 2 * T;
  int q = 0, r = n;   // |n| is a few periods at most (window / clip length): compare-and-step beats an integer division
  while (r < 0) {
    r += two;
    --q;
  }
  while (r >= two) {
    r -= two;
    ++q;
  }
  const double p2 = r <= T ? P[r] : 2.0 * P[T] - P[two - r];
  return (double)q * (2.0 * P[T]) + p2;
}
__global__ __launch_bounds__(256) void cmvnw_tile_kernel(const float* __restrict__ src, int max_frames, int ncols,
                                                         const int32_t* __restrict__ n_frames, int win, int variance, int cg,
                                                         float* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_cw[];
  const int utt = blockIdx.y, c0 = blockIdx.x * cg;
  const int nc = ncols - c0 < cg ? ncols - c0 : cg;
  int T = n_frames ? n_frames[utt] : max_frames;
  T = T < max_frames ? T : max_frames;
  if (T <= 0 || nc <= 0) return;
  double* P = reinterpret_cast<double*>(smem_cw);           // [nc][T + 1]
  double* Q = P + (size_t)cg * (max_frames + 1);            // [nc][T + 1]
  float* A = reinterpret_cast<float*>(Q + (size_t)cg * (max_frames + 1));   // [nc][T], column-major
  const float* base = src + (int64_t)utt * max_frames * ncols + c0;
  float* out = dst + (int64_t)utt * max_frames * ncols + c0;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int half = (win - 1) / 2;
  const double inv_win = 1.0 / (double)win;
  for (int idx = t; idx < nc * T; idx += 256) {
    const int r = idx / nc, c = idx - r * nc;
    A[c * T + r] = base[(int64_t)r * ncols + c];
  }
  __syncthreads();
  // prefix sums of A (and of A * A) along time: a wave per column, 64 rows per scan step
  auto scan_columns = [&](bool squares) {
    for (int c = wave; c < nc; c += 4) {
      double carry = 0.0, carry2 = 0.0;
      double* Pc = P + (size_t)c * (T + 1);
      double* Qc = Q + (size_t)c * (T + 1);
      if (lane == 0) {
        Pc[0] = 0.0;
        if (squares) Qc[0] = 0.0;
      }
      for (int s0 = 0; s0 < T; s0 += 64) {
        const double v = s0 + lane < T ? (double)A[c * T + s0 + lane] : 0.0;
        double a = v, b = v * v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const double ua = __shfl_up(a, d, 64), ub = __shfl_up(b, d, 64);
          if (lane >= d) {
            a += ua;
            b += ub;
          }
        }
        if (s0 + lane < T) {
          Pc[s0 + lane + 1] = carry + a;
          if (squares) Qc[s0 + lane + 1] = carry2 + b;
        }
        carry += __shfl(a, 63, 64);
        carry2 += __shfl(b, 63, 64);
      }
    }
  };
  scan_columns(false);
  __syncthreads();
  for (int idx = t; idx < nc * T; idx += 256) {
    const int r = idx / nc, c = idx - r * nc;
    const double* Pc = P + (size_t)c * (T + 1);
    const double mean = (cw_prefix_at(Pc, T, r + half + 1) - cw_prefix_at(Pc, T, r - half)) * inv_win;
    const float y = (float)((double)A[c * T + r] - mean);
    if (variance) A[c * T + r] = y;
    else out[(int64_t)r * ncols + c] = y;
  }
  if (!variance) return;
  __syncthreads();
  scan_columns(true);
  __syncthreads();
  for (int idx = t; idx < nc * T; idx += 256) {
    const int r = idx / nc, c = idx - r * nc;
    const double* Pc = P + (size_t)c * (T + 1);
    const double* Qc = Q + (size_t)c * (T + 1);
    const double mean = (cw_prefix_at(Pc, T, r + half + 1) - cw_prefix_at(Pc, T, r - half)) * inv_win;
    double var = (cw_prefix_at(Qc, T, r + half + 1) - cw_prefix_at(Qc, T, r - half)) * inv_win - mean * mean;
    var = var > 0.0 ? var : 0.0;
    out[(int64_t)r * ncols + c] = (float)((double)A[c * T + r] / (sqrt(var) + 9.313225746154785e-10));
  }
}

// ---- 'derivative' features, bug-compatible (Q11) -------------------------------------------------
__global__ __launch_bounds__(256) void derivative_kernel(const float* __restrict__ in, int64_t total, int ncols,
                                                         int delta, float inv_scale, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    const float* row = in + r * ncols;
    float acc = 0.f;
    for (int k = 1; k <= delta; ++k) acc += (float)k * row[min(c + k, ncols - 1)];
    out[i] = acc * inv_scale;
  }
}

// ---- log power spectrum ---------------------------------------------------------------------------
// order-preserving float <-> uint map so that atomicMax on the integers is a max on the floats
__device__ __forceinline__ unsigned flip(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unflip(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__global__ __launch_bounds__(256) void log_power_kernel(float* __restrict__ p, int64_t n, unsigned* __restrict__ gmax) {
  float m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = p[i];
    v = v <= 1e-20f ? 1e-20f : v;            // processing.py:192
    v = 10.0f * log10f(v);
    p[i] = v;
    m = fmaxf(m, v);
  }
  if (gmax) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(gmax, flip(m));
  }
}
__global__ __launch_bounds__(256) void sub_max_kernel(float* __restrict__ p, int64_t n, const unsigned* __restrict__ gmax) {
  const float m = unflip(*gmax);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] -= m;
}

// ---- feature cube: each crop is one contiguous run of crop_frames * ncols floats ---------------
__global__ __launch_bounds__(256) void cube_gather_kernel(const float* __restrict__ feat, int max_frames, int ncols,
                                                          const int32_t* __restrict__ crop, int n_crops,
                                                          int crop_frames, int64_t n_jobs, const double* __restrict__ stats,
                                                          float* __restrict__ out) {
  // stats (svk_cube_gather_cmvn): [utt][2][ncols] mean and 1 / (std + 2^-30) of svk_cmvn_stats, applied to every copied value
  // with cmvn_kernel's own expression -- the cube of normalised features without the pass that normalises ALL rows in place
  const int run = crop_frames * ncols;
  for (int64_t job = blockIdx.x; job < n_jobs; job += gridDim.x) {
    const int64_t utt = job / n_crops;
    int start = crop[job];
    const bool none = start < 0;  // svk_cube_draw_crops marks clips that are too short
    start = none ? max_frames : (start > max_frames ? max_frames : start);
    const float* src = feat + (utt * max_frames + start) * ncols;
    float* dst = out + job * run;
    const int avail = (max_frames - start) * ncols;  // never read past the clip's rows
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                     (run & 3) == 0 && avail >= run;
    if (stats && !none) {
      const double* st = stats + utt * 2 * ncols;
      if (vec && (ncols & 3) == 0) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
        f32x4* d4 = reinterpret_cast<f32x4*>(dst);
        for (int i = threadIdx.x; i < run / 4; i += blockDim.x) {
          const int c = (4 * i) % ncols;     // a 16-byte piece stays inside one row: ncols is a multiple of 4
          const f32x4 v = s4[i];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (float)(((double)v[e] - st[c + e]) * st[ncols + c + e]);
          d4[i] = o;
        }
      } else {
        for (int i = threadIdx.x; i < run; i += blockDim.x) {
          const int c = i % ncols;
          dst[i] = i < avail ? (float)(((double)src[i] - st[c]) * st[ncols + c]) : 0.f;
        }
      }
    } else if (vec) {
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
      f32x4* d4 = reinterpret_cast<f32x4*>(dst);
      for (int i = threadIdx.x; i < run / 4; i += blockDim.x) d4[i] = s4[i];
    } else {
      for (int i = threadIdx.x; i < run; i += blockDim.x) dst[i] = i < avail ? src[i] : 0.f;
    }
  }
}

// ---- crop starts drawn on device ---------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void draw_crops_kernel(const int32_t* __restrict__ n_frames, int n_utt,
                                                         int64_t first_utt, const int64_t* __restrict__ utt_index,
                                                         int n_crops, int crop_frames,
                                                         uint64_t seed, int32_t* __restrict__ crop,
                                                         int32_t* __restrict__ bad) {
  const int64_t total = (int64_t)n_utt * n_crops;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int u = (int)(i / n_crops), c = (int)(i - (int64_t)u * n_crops);
    const int range = n_frames[u] - crop_frames;
    int v = -1;
    if (range > 0) {
      const uint64_t gid = utt_index ? (uint64_t)utt_index[u] : (uint64_t)(first_utt + u);
      const uint64_t r = splitmix64(splitmix64(seed ^ gid) + (uint64_t)c);
      v = (int)__umul64hi(r, (uint64_t)range);  // floor(r / 2^64 * range)
    } else if (c == 0 && bad) {
      atomicAdd(bad, 1);
    }
    crop[i] = v;
  }
}

// twiddles for spectrum_fft_kernel: tw1 [8][64] | tw2 [8][64] | tw3 [5][64]
cplx* g_spectrum_tables[64] = {nullptr};

int spectrum_tables(svk_ctx* ctx, const cplx** out) {
  if (ctx->device < 0 || ctx->device >= 64) return svk_fail(ctx, SVK_ERR_BAD_ARG, "device index out of range");
  cplx*& d = g_spectrum_tables[ctx->device];
  if (!d) {
    const double PI = 3.14159265358979323846;
    static cplx h[1024 + 5 * 64];
    for (int r = 0; r < 8; ++r)
      for (int l = 0; l < 64; ++l) {
        const double a1 = -2.0 * PI * (double)(l * r) / 512.0, a2 = -2.0 * PI * (double)((l & 7) * r) / 64.0;
        h[r * 64 + l] = mk((float)cos(a1), (float)sin(a1));
        h[512 + r * 64 + l] = mk((float)cos(a2), (float)sin(a2));
      }
    for (int q = 0; q < 5; ++q)
      for (int l = 0; l < 64; ++l) {
        const double a = -2.0 * PI * (double)(l + 64 * q) / 1024.0;
        h[1024 + q * 64 + l] = mk((float)cos(a), (float)sin(a));
      }
    SVK_HIP(ctx, hipSetDevice(ctx->device));
    SVK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d), sizeof(h)));
    SVK_HIP(ctx, hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  }
  *out = d;
  return SVK_OK;
}

}  // namespace

extern "C" {

int svk_preemphasis(svk_ctx* ctx, const void* d_in, int pcm_dtype, int64_t n, int32_t shift, float cof,
                    float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n >= 0, "n negative");
  if (n == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, pcm_dtype == SVK_PCM_I16 || pcm_dtype == SVK_PCM_F32, "pcm_dtype");
  int64_t sm = (int64_t)shift % n;
  if (sm < 0) sm += n;
  const unsigned grid = capped_grid(ctx, n, 256);
  if (pcm_dtype == SVK_PCM_I16)
    hipLaunchKernelGGL(preemph_kernel<int16_t>, dim3(grid), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const int16_t*>(d_in), n, sm, cof, d_out);
  else
    hipLaunchKernelGGL(preemph_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float*>(d_in), n, sm, cof, d_out);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_stack_frames(svk_ctx* ctx, const float* d_sig, int64_t n, int32_t frame_len, int32_t stride,
                     int32_t n_frames, const float* d_window, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n >= 0 && frame_len >= 1 && stride >= 1 && n_frames >= 0, "negative or zero geometry");
  if (n_frames == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_sig && d_out, "NULL buffer");
  const int64_t total = (int64_t)n_frames * frame_len;
  hipLaunchKernelGGL(stack_frames_kernel, dim3(capped_grid(ctx, total, 256)), dim3(256), 0, ctx->stream, d_sig, n,
                     frame_len, stride, total, d_window, d_out);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_spectrum(svk_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_len, int32_t nfft,
                 int32_t power, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_frames >= 0 && frame_len >= 1 && nfft >= 2, "geometry");
  if (n_frames == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_frames && d_out, "NULL buffer");
  if (nfft == 512 || nfft == 1024) {
    const cplx* tw = nullptr;
    int rc = spectrum_tables(ctx, &tw);
    if (rc != SVK_OK) return rc;
    if (nfft == 512) {
      const unsigned grid = capped_grid(ctx, (n_frames + 1) / 2, SPEC_WAVES);
      hipLaunchKernelGGL(spectrum_fft_kernel<false>, dim3(grid), dim3(64 * SPEC_WAVES), 0, ctx->stream, d_frames,
                         n_frames, frame_len, power, tw, d_out);
    } else {
      const unsigned grid = capped_grid(ctx, n_frames, SPEC_WAVES);
      hipLaunchKernelGGL(spectrum_fft_kernel<true>, dim3(grid), dim3(64 * SPEC_WAVES), 0, ctx->stream, d_frames,
                         n_frames, frame_len, power, tw, d_out);
    }
  } else {
    const bool pow2 = (nfft & (nfft - 1)) == 0 && nfft >= 4 && nfft <= 8192;
    const int feff = frame_len < nfft ? frame_len : nfft;
    const int count = pow2 ? nfft / 2 : nfft;
    const size_t lds = pow2 ? sizeof(cplx) * ((size_t)nfft / 2 + (size_t)(256 / std::max(1, std::min(256, nfft / 4))) * nfft)
                            : sizeof(cplx) * (size_t)nfft + sizeof(float) * (size_t)DFT_FR * feff;
    if (lds > (size_t)ctx->lds_per_cu)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "fft_points %d with %d-sample frames needs %zu bytes of LDS (limit %d)", nfft,
                      frame_len, lds, ctx->lds_per_cu);
    int rc = svk_ensure_work(ctx, sizeof(cplx) * (size_t)count);
    if (rc != SVK_OK) return rc;
    cplx* tw = reinterpret_cast<cplx*>(ctx->work);
    hipLaunchKernelGGL(twiddle_table_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, tw, nfft, count);
    SVK_LAUNCH_CHECK(ctx);
    if (pow2) {
      const int fpb = 256 / std::max(1, std::min(256, nfft / 4));
      SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(spectrum_pow2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(spectrum_pow2_kernel, dim3(capped_grid(ctx, n_frames, fpb)), dim3(256), lds, ctx->stream, d_frames,
                         n_frames, frame_len, nfft, power, tw, d_out);
    } else {
      SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(spectrum_dft_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(spectrum_dft_kernel, dim3(capped_grid(ctx, n_frames, DFT_FR)), dim3(256), lds, ctx->stream, d_frames,
                         n_frames, frame_len, nfft, power, tw, d_out);
    }
  }
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

// svk_cmvn (d_stats_out == NULL: normalise in place) and svk_cmvn_stats (statistics only) share the two paths
static int cmvn_launch(svk_ctx* ctx, float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                       const int32_t* d_n_frames, int32_t variance, double* d_stats_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 0 && n_cols >= 0, "negative shape");
  if (n_utt == 0 || max_frames == 0 || n_cols == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_feat, "NULL buffer");
  // long clips: chunks of a clip go to separate workgroups (see cmvn_partial_kernel); SVK_CMVN_SPLIT=0 / 1 forces a path
  const char* force = getenv("SVK_CMVN_SPLIT");
  const bool split = force ? force[0] == '1' : max_frames > 1024;
  if (split && n_utt <= 65535) {
    const int n_chunks = (max_frames + CMVN_CHUNK - 1) / CMVN_CHUNK;
    const size_t part_bytes = sizeof(double) * (size_t)n_utt * n_chunks * 2 * n_cols;
    const size_t stats_bytes = sizeof(double) * (size_t)n_utt * 2 * n_cols;
    const int rc = svk_ensure_work(ctx, part_bytes + stats_bytes);
    if (rc != SVK_OK) return rc;
    double* part = reinterpret_cast<double*>(ctx->work);
    double* stats = d_stats_out ? d_stats_out : part + (size_t)n_utt * n_chunks * 2 * n_cols;
    hipLaunchKernelGGL(cmvn_partial_kernel, dim3(n_chunks, n_utt), dim3(256), 0, ctx->stream, d_feat, max_frames, n_cols,
                       d_n_frames, n_chunks, part);
    SVK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(cmvn_stats_kernel, dim3(n_utt), dim3(256), 0, ctx->stream, part, max_frames, n_cols, d_n_frames, n_chunks,
                       variance, stats);
    SVK_LAUNCH_CHECK(ctx);
    if (!d_stats_out) {
      hipLaunchKernelGGL(cmvn_apply_kernel, dim3(n_chunks, n_utt), dim3(256), 0, ctx->stream, d_feat, max_frames, n_cols,
                         d_n_frames, stats);
      SVK_LAUNCH_CHECK(ctx);
    }
    return SVK_OK;
  }
  if (d_stats_out)
    hipLaunchKernelGGL(cmvn_kernel<false>, dim3(n_utt), dim3(256), 0, ctx->stream, d_feat, max_frames, n_cols, d_n_frames,
                       variance, d_stats_out);
  else
    hipLaunchKernelGGL(cmvn_kernel<true>, dim3(n_utt), dim3(256), 0, ctx->stream, d_feat, max_frames, n_cols, d_n_frames,
                       variance, (double*)nullptr);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_cmvn(svk_ctx* ctx, float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
             const int32_t* d_n_frames, int32_t variance) {
  return cmvn_launch(ctx, d_feat, n_utt, max_frames, n_cols, d_n_frames, variance, nullptr);
}

int svk_cmvn_stats(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                   const int32_t* d_n_frames, int32_t variance, double* d_stats) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, d_stats || n_utt == 0 || n_cols == 0 || max_frames == 0, "d_stats is NULL");
  return cmvn_launch(ctx, const_cast<float*>(d_feat), n_utt, max_frames, n_cols, d_n_frames, variance, d_stats);
}

int svk_mel_features(svk_ctx* ctx, const float* d_power, int32_t n_frames, int32_t n_bins, const float* d_bank,
                     int32_t num_filters, int32_t out_kind, int32_t num_ceps, int32_t dc_elimination, float* d_feat,
                     float* d_energy) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_frames >= 0 && n_bins >= 1 && num_filters >= 1, "shape");
  SVK_REQUIRE(ctx, out_kind >= SVK_OUT_MFE && out_kind <= SVK_OUT_MFCC, "out_kind");
  if (num_filters > 1024) return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "at most 1024 filters, got %d", num_filters);
  if (out_kind == SVK_OUT_MFCC) SVK_REQUIRE(ctx, num_ceps >= 1 && num_ceps <= num_filters, "1 <= num_ceps <= num_filters");
  if (n_frames == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_power && d_bank && d_feat, "NULL buffer");
  const float* dct = nullptr;
  if (out_kind == SVK_OUT_MFCC) {  // DCT-II rows, rebuilt per call in the handle's workspace (stream-ordered)
    const int n = num_ceps * num_filters;
    int rc = svk_ensure_work(ctx, sizeof(float) * (size_t)n);
    if (rc != SVK_OK) return rc;
    hipLaunchKernelGGL(dct_table_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                       reinterpret_cast<float*>(ctx->work), num_ceps, num_filters);
    SVK_LAUNCH_CHECK(ctx);
    dct = reinterpret_cast<const float*>(ctx->work);
  }
  if (num_filters <= 256) {   // the mel product on MFMA
    const size_t lds = sizeof(float) * (size_t)(MM_FR * MM_LD + 64 * MM_LD + MM_FR + MM_FR * num_filters);
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mel_features_mfma_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mel_features_mfma_kernel, dim3((unsigned)((n_frames + MM_FR - 1) / MM_FR)), dim3(256), lds, ctx->stream,
                       d_power, n_frames, n_bins, d_bank, num_filters, out_kind, num_ceps, dc_elimination, dct, d_feat,
                       d_energy);
    SVK_LAUNCH_CHECK(ctx);
    return SVK_OK;
  }
  const size_t lds = sizeof(float) * (size_t)(MEL_FR * MEL_CH + 64 * (MEL_CH + 1) + MEL_FR + MEL_FR * num_filters);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mel_features_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mel_features_kernel, dim3((unsigned)((n_frames + MEL_FR - 1) / MEL_FR)), dim3(256), lds, ctx->stream,
                     d_power, n_frames, n_bins, d_bank, num_filters, out_kind, num_ceps, dc_elimination, dct, d_feat,
                     d_energy);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_cmvnw(svk_ctx* ctx, const float* d_in, int32_t n_utt, int32_t max_frames, int32_t n_cols,
              const int32_t* d_n_frames, int32_t win, int32_t variance, float* d_tmp, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 0 && n_cols >= 0, "negative shape");
  SVK_REQUIRE(ctx, win >= 1 && (win & 1) == 1, "win must be odd");
  if (n_utt == 0 || max_frames == 0 || n_cols == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_out && d_in != d_out, "NULL or aliased buffer");
  SVK_REQUIRE(ctx, !variance || (d_tmp && d_tmp != d_in && d_tmp != d_out), "variance pass needs a distinct d_tmp");
  if (max_frames <= CW_CAP) {   // the clip's columns fit LDS: prefix sums, both passes in one kernel
    const int cg = std::min<int>(n_cols, CW_CAP / max_frames);
    const size_t lds = (size_t)cg * ((size_t)max_frames * sizeof(float) + 2 * (size_t)(max_frames + 1) * sizeof(double));
    SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cmvnw_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(cmvnw_tile_kernel, dim3((unsigned)((n_cols + cg - 1) / cg), (unsigned)n_utt), dim3(256), lds, ctx->stream,
                       d_in, max_frames, n_cols, d_n_frames, win, variance, cg, d_out);
    SVK_LAUNCH_CHECK(ctx);
    return SVK_OK;
  }
  // rows per thread: long enough to amortise the direct window sum of a segment's first row, short enough
  // to keep a few hundred threads per clip
  const int seg = max_frames <= 1024 ? 32 : 128;
  const int64_t per = (int64_t)((max_frames + seg - 1) / seg) * n_cols;
  const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((per + 255) / 256, ctx->num_cu * 4)), (unsigned)n_utt);
  hipLaunchKernelGGL(cmvnw_kernel, grid, dim3(256), 0, ctx->stream, d_in, max_frames, n_cols, d_n_frames, win, 0, seg,
                     variance ? d_tmp : d_out);
  SVK_LAUNCH_CHECK(ctx);
  if (variance) {
    hipLaunchKernelGGL(cmvnw_kernel, grid, dim3(256), 0, ctx->stream, d_tmp, max_frames, n_cols, d_n_frames, win, 1, seg,
                       d_out);
    SVK_LAUNCH_CHECK(ctx);
  }
  return SVK_OK;
}

int svk_derivative(svk_ctx* ctx, const float* d_in, int64_t n_rows, int32_t n_cols, int32_t delta, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_rows >= 0 && n_cols >= 0 && delta >= 1, "shape / delta");
  const int64_t total = n_rows * n_cols;
  if (total == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_out && d_in != d_out, "NULL or aliased buffer");
  double scale = 0.0;
  for (int k = 1; k <= delta; ++k) scale += 2.0 * k * k;  // processing.py:233
  hipLaunchKernelGGL(derivative_kernel, dim3(capped_grid(ctx, total, 256)), dim3(256), 0, ctx->stream, d_in, total,
                     n_cols, delta, (float)(1.0 / scale), d_out);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_log_power(svk_ctx* ctx, float* d_power, int64_t n, int32_t normalize) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n >= 0, "n negative");
  if (n == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_power, "NULL buffer");
  unsigned* gmax = normalize ? reinterpret_cast<unsigned*>(ctx->scratch) : nullptr;
  if (gmax) SVK_HIP(ctx, hipMemsetAsync(gmax, 0, sizeof(unsigned), ctx->stream));  // flip() of anything is > 0
  hipLaunchKernelGGL(log_power_kernel, dim3(capped_grid(ctx, n, 256)), dim3(256), 0, ctx->stream, d_power, n, gmax);
  SVK_LAUNCH_CHECK(ctx);
  if (gmax) {
    hipLaunchKernelGGL(sub_max_kernel, dim3(capped_grid(ctx, n, 256)), dim3(256), 0, ctx->stream, d_power, n, gmax);
    SVK_LAUNCH_CHECK(ctx);
  }
  return SVK_OK;
}

int svk_cube_draw_crops(svk_ctx* ctx, const int32_t* d_n_frames, int32_t n_utt, int64_t first_utt,
                        const int64_t* d_utt_index, int32_t n_crops, int32_t crop_frames, uint64_t seed,
                        int32_t* d_crop_idx, int32_t* d_bad_count) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && n_crops >= 0 && crop_frames >= 0, "negative shape");
  const int64_t total = (int64_t)n_utt * n_crops;
  if (total == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_n_frames && d_crop_idx, "NULL buffer");
  hipLaunchKernelGGL(draw_crops_kernel, dim3(capped_grid(ctx, total, 256)), dim3(256), 0, ctx->stream, d_n_frames,
                     n_utt, first_utt, d_utt_index, n_crops, crop_frames, seed, d_crop_idx, d_bad_count);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

static int cube_gather_launch(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                              const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const double* d_stats, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && n_crops >= 0 && crop_frames >= 0 && n_cols >= 0 && max_frames >= 0, "negative shape");
  const int64_t jobs = (int64_t)n_utt * n_crops;
  if (jobs == 0 || crop_frames == 0 || n_cols == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_feat && d_crop_idx && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, crop_frames <= max_frames, "crop_frames exceeds max_frames");
  const unsigned grid = (unsigned)std::min<int64_t>(jobs, (int64_t)ctx->num_cu * 16);
  hipLaunchKernelGGL(cube_gather_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_feat, max_frames, n_cols,
                     d_crop_idx, n_crops, crop_frames, jobs, d_stats, d_out);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

int svk_cube_gather(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                    const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, float* d_out) {
  return cube_gather_launch(ctx, d_feat, n_utt, max_frames, n_cols, d_crop_idx, n_crops, crop_frames, nullptr, d_out);
}

int svk_cube_gather_cmvn(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                         const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const double* d_stats, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, d_stats || n_utt == 0, "d_stats is NULL");
  return cube_gather_launch(ctx, d_feat, n_utt, max_frames, n_cols, d_crop_idx, n_crops, crop_frames, d_stats, d_out);
}

}  // extern "C"
