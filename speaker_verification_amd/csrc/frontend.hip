// Fused front end for gfx950: PCM -> [pre-emphasis] -> frames -> rFFT power spectrum ->
// frame energy -> mel filterbank -> [log] -> [DCT-II] -> [c0 := log E].
//
// Replaces, per clip, /root/reference/speech_feature_extraction/speechpy/
//   processing.py:45-58 (preemphasis, circular)   processing.py:112-138 (framing, no padding)
//   processing.py:142-174 (|rfft|^2 / nfft)       feature.py:202-205 (frame energy, 0 -> eps)
//   feature.py:216-217 (mel projection, 0 -> eps) feature.py:146-153 (log, DCT-II ortho, c0 := log E)
//
// Work decomposition: one workgroup per CU, as many waves as LDS allows (up to 12); the
// workgroup copies the filterbank / DCT operand fragments into LDS once (they are re-read
// for every tile: fetching them from L2 per tile cost 4x the PCM traffic and its latency),
// then every wave works on its own: it owns a TILE of 8 (or 16) consecutive frames of one
// clip, a private LDS slice, and loops over tiles with no further workgroup barrier (the waves
// of a workgroup draw the workgroup's tiles from a counter in LDS).  The two standard
// configurations run instances with their framing and output shape fixed at compile time (Spec).
//   1. the tile's sample span is read from HBM once with 16-byte loads issued back to back and
//      parked in LDS -- int16 clips as raw int16 (pre-emphasis x[n] - c x[n-1] is applied when a
//      frame is read), float clips or an unusual pre-emphasis shift as pre-emphasised f32;
//   2. 512-point complex FFTs run across the wave, two per trip pipelined through one scratch:
//      8 points per lane, three radix-8 passes in registers on packed-f32 instructions
//      (fft_wave.h), two transposes through a padded (bank-conflict-free) LDS scratch.  nfft = 512 packs TWO real frames into one
//      complex FFT; nfft = 1024 packs the even/odd samples of ONE frame.  The untangle step
//      pairs bin k with bin N-k, which lives in lane 64-l: one wave shuffle per register, no
//      LDS, and only for the bins the mel filters read;
//   3. those power bins go to a [TILE x KP] LDS tile; the frame energy (all bins) comes from
//      Parseval's identity: per-lane partial sums, three DPP adds to 8 group sums parked behind the
//      frame's power bins, finished by two MFMAs against ones -- and only when something consumes it;
//   4. mel energies^T = filterbank x P^T on v_mfma_f32_16x16x4_f32 (exact f32), skipping
//      the 16-bin chunks where a 16-filter tile is identically zero (Q2: the bank is
//      ~97 % zeros);  log;  the accumulator layout of that product is exactly the B
//      operand layout of the next one, so cepstra^T = DCT x log(mel)^T follows with no
//      data movement;
//   5. results go straight to HBM.
// (v_mfma_f32_4x4x1_16b_f32, which could follow the bank's sparsity filter group by filter
// group, was measured at 15 cycles per instruction = half the MAC rate of 16x16x4:
// tools/probes/mfma4x4_probe.hip -- not worth it.)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "fft_wave.h"
#include "svk_internal.h"

using namespace svk_fft;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RAW_OFF = 8;    // int16 staging: samples start 16 bytes in, slot RAW_OFF-1 holds the sample before the span
constexpr int MAX_FT = 4;    // 16-filter tiles  (<= 64 filters)
constexpr int PT_PAD = 8;    // ptile row padding (floats): conflict-free ds_read_b128; also holds the frame's 8 energy group sums
constexpr float EPS64 = 2.220446049250313e-16f;  // np.finfo(float).eps, functions.py:62

// Phase-ablation switches for tuning: compiled in only with -DSVK_TUNING
// (`make TUNING=1`), selected at run time by SVK_FE_ABLATE = 1 skip staging | 2 skip the FFT loop |
// 4 skip mel..output | 8 skip the mel MFMAs | 16 skip the DCT | 32 skip log..output | 64 skip zero fill.
#ifdef SVK_TUNING
#define SVK_ABLATE(p, bit) (((p).ablate & (bit)) != 0)
#else
#define SVK_ABLATE(p, bit) false
#endif

struct FrontendParams {
  const void* pcm;
  const int64_t* offsets;
  const int32_t* lengths;
  int64_t clip_stride;
  int32_t clip_len;
  // Gathered input (int16 PCM only): a clip's signal is the concatenation of CHUNKS of `chunk` samples of its PCM, chunk q of
  // clip u = PCM chunk src_chunk[u * chunk_stride + q] -- the index form of svk_vad_energy's compaction (d_src_frame): the
  // kept frames are read where they lie instead of being copied to the front of a second buffer first.  NULL = plain clips.
  const int32_t* src_chunk;
  int32_t chunk, chunk_stride;
  float chunk_inv;   // 1 / chunk: sample index -> chunk without an integer division (indices stay below 2^24: exact in float)
  int32_t n_utt, max_frames, tiles_per_utt;
  int32_t flen, flen_eff, stride, nfilt, ncols, out_kind, dc_elim, preemph, pre_shift;
  float pre_cof;
  float escale;      // input_scale^2: applied to the frame energy (the mel path carries it in the bank weights)
  int32_t kp;        // power bins kept for the mel product, multiple of 16
  int32_t sig_bytes; // LDS bytes reserved for the staged samples (multiple of 16)
  int32_t n_ft, n_ct;
  int32_t chunk_lo[MAX_FT], chunk_hi[MAX_FT];
  const cplx* tw1;    // [8][64]  W512^(lane*r)
  const cplx* tw2;    // [8][64]  W64^((lane&7)*r)
  const cplx* tw3;    // [5][64]  W1024^(lane+64q), q<4; [4][0] = W1024^256
  const f32x4* fbfrag;  // [n_slots][64]: only the non-zero (filter tile, 16-bin chunk) blocks
  const float* dctfrag; // [n_ct][n_ft][4][64]
  int32_t n_slots, slot_base[MAX_FT];  // block (t, u) lives in slot slot_base[t] + u - chunk_lo[t]
  int32_t table_bytes, wave_bytes;     // LDS: shared tables, then one slice per wave
  int32_t need_energy; // frame energies are consumed (c0 := log E, or d_energy given)
  int32_t n_steps; // register steps a frame reaches: ceil(flen_eff / 64), or / 128 for nfft 1024
  int32_t ablate;  // tuning only (SVK_FE_ABLATE): 1 skip staging, 2 skip the FFT loop, 4 skip mel/DCT/output
  float* feat;
  float* energy;
  int32_t* n_frames;
};

template <typename PcmT>
__device__ __forceinline__ void stage_span(const FrontendParams& p, const PcmT* x, int64_t s0, int need,
                                           int len, float* sig, int lane) {
  constexpr int V = 16 / (int)sizeof(PcmT);
  typedef PcmT vec_t __attribute__((ext_vector_type(V)));
  const bool fast = (!p.preemph || p.pre_shift == 1) && ((reinterpret_cast<uintptr_t>(x + s0) & 15) == 0);
  for (int i = lane * V; i < need; i += 64 * V) {
    const int64_t idx = s0 + i;
    float v[V];
    float o[V];
    if (fast && idx + V <= len) {
      vec_t raw = *reinterpret_cast<const vec_t*>(x + idx);
#pragma unroll
      for (int e = 0; e < V; ++e) v[e] = (float)raw[e];
      if (p.preemph) {
        const float prev = (float)x[idx == 0 ? len - 1 : idx - 1];
        o[0] = v[0] - p.pre_cof * prev;
#pragma unroll
        for (int e = 1; e < V; ++e) o[e] = v[e] - p.pre_cof * v[e - 1];
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = v[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const int64_t id = idx + e;
        float val = 0.f;
        if (id < len) {
          val = (float)x[id];
          if (p.preemph) {
            int64_t j = (id - p.pre_shift) % len;
            if (j < 0) j += len;
            val -= p.pre_cof * (float)x[j];
          }
        }
        o[e] = val;
      }
    }
#pragma unroll
    for (int e = 0; e < V; e += 4)
      *reinterpret_cast<f32x4*>(sig + i + e) = (f32x4){o[e], o[e + 1], o[e + 2], o[e + 3]};
  }
}

// int16 clips with no pre-emphasis or the usual shift of 1: park the RAW samples (2 bytes each,
// half the LDS of the f32 form) and pre-emphasise when a frame is read: x[n] - c * x[n-1] needs
// one extra sample in front of the span (the clip's LAST sample when the span starts the clip:
// np.roll wraps, Q5).
// `tab` (gathered input, FrontendParams::src_chunk): sample j of the clip's signal = x[tab[j / chunk] * chunk + j % chunk]; chunk is a
// multiple of 8 and every span starts on a multiple of 8, so a 16-byte vector never straddles two chunks.
// q = j / chunk from a float reciprocal, corrected by one step either way (an integer division is ~40 vector instructions, four
// of them per lane and tile were 8 % of this issue-bound kernel): j < 2^24 (the host checks), so (float)j is exact and the
// estimate is off by at most one.
__device__ __forceinline__ int64_t gathered_index(const int32_t* tab, int chunk, float chunk_inv, int64_t j) {
  const int ji = (int)j;
  int q = (int)((float)ji * chunk_inv);
  int r = ji - q * chunk;
  if (r < 0) { --q; r += chunk; }
  if (r >= chunk) { ++q; r -= chunk; }
  return (int64_t)tab[q] * chunk + r;
}

__device__ __forceinline__ void stage_raw16(bool pre, const int16_t* x, int64_t s0, int need, int len, int16_t* sigh,
                                            int lane, const int32_t* tab = nullptr, int chunk = 0, float chunk_inv = 0.f) {
  typedef short vec_t __attribute__((ext_vector_type(8)));
  constexpr int NV = 3;  // 16-byte loads in flight per lane: an 8-frame tile at a 160-sample hop is 3 x 512 samples
  if (lane == 0 && pre) {
    const int64_t jp = s0 == 0 ? (int64_t)len - 1 : s0 - 1;
    sigh[RAW_OFF - 1] = x[tab ? gathered_index(tab, chunk, chunk_inv, jp) : jp];
  }
  // wave-uniform: the span starts on a 16-byte boundary and its last vector ends inside the clip
  // (every tile of a clip but possibly the last one), so no load needs patching
  const bool whole = (tab ? (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (s0 & 7) == 0
                          : (reinterpret_cast<uintptr_t>(x + s0) & 15) == 0) && s0 + ((need + 7) & ~7) <= len;
  if (whole) {
    for (int base = 0; base < need; base += NV * 512) {
      vec_t raw[NV];
      // all loads first (a load per loop trip followed by its own wait cost one HBM round trip EACH) ...
      if (tab) {
        int64_t src[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int i = base + k * 512 + lane * 8;
          src[k] = gathered_index(tab, chunk, chunk_inv, s0 + (i < need ? i : 0));   // (the table entries first: dependent loads)
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) raw[k] = *reinterpret_cast<const vec_t*>(x + src[k]);
      } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const int i = base + k * 512 + lane * 8;
          raw[k] = *reinterpret_cast<const vec_t*>(x + s0 + (i < need ? i : 0));  // clamped: always a valid aligned address
        }
      }
      // ... then the LDS writes
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int i = base + k * 512 + lane * 8;
        if (i < need) *reinterpret_cast<vec_t*>(sigh + RAW_OFF + i) = raw[k];
      }
    }
  } else {
    for (int i = lane * 8; i < need; i += 64 * 8) {
      vec_t raw;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int64_t j = s0 + i + e;
        raw[e] = j < len ? x[tab ? gathered_index(tab, chunk, chunk_inv, j) : j] : (short)0;
      }
      *reinterpret_cast<vec_t*>(sigh + RAW_OFF + i) = raw;
    }
  }
}

// Sums of two values over each group of 8 consecutive lanes, DPP adds only: an inclusive scan inside
// the 16-lane rows (row_shr 1, 2, 4; lanes shifted in from outside the row read 0) leaves the sum of
// lanes 8 m .. 8 m + 7 in lane 8 m + 7.  Written as v_add_f32_dpp (add and lane move in ONE
// instruction; the update_dpp builtin costs a zero-init, a move and an add per step); a VALU result
// needs two wait states before a DPP read, which the other value's step plus one s_nop provide.
// The 8 group sums of a frame are finished by the matrix cores (see the energy MFMA below).
__device__ __forceinline__ void group8_sum2(float& a, float& b) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b));
}

// Frame samples -> the 8 complex registers of a lane.  NSTEPS = how many of the 8 register steps
// the frame reaches (frame_len / 64, or / 128 for the even-odd packing, rounded up); steps
// before the last are read unconditionally, the last one clamps its address and masks, the
// rest are the zero padding up to nfft (Q6).  Straight-line code: per-lane `if (n < frame_len)`
// around every read cost 28 exec-mask branches and as many serialised LDS round trips per FFT.
typedef const volatile __attribute__((address_space(3))) int16_t* lds_i16p;  // volatile needs the LDS address space spelled out
typedef const volatile __attribute__((address_space(3))) int* lds_i32p;
template <bool SPLIT1024, bool RAW16>
struct FrameReader {
  const float* sig;
  const int16_t* sigh;
  int off_a, off_b, flen;
  float cof;
  bool hasb;
  int lane;

  // PRE is a template parameter, not a select: with the LDS read of x[n-1] inside a `pre ? :`
  // arm the compiler emitted a branch per sample, each read followed by its own s_waitcnt
  // (20 serialised LDS round trips per FFT); straight-line code batches the reads.  The raw
  // samples are read through a volatile pointer so that x[n-1], x[n] are NOT merged into one
  // 2-byte-misaligned ds_read_b32 (measured: the merged form made the nfft-1024 kernel 16 % slower).
  template <bool PRE>
  __device__ __forceinline__ float sample(int off, int idx) const {
    if constexpr (RAW16) {
      const lds_i16p s = (lds_i16p)(sigh + RAW_OFF + off + idx);
      const float x0 = (float)s[0];
      if constexpr (PRE) return x0 - cof * (float)s[-1];  // x[n] - c x[n-1] (shift 1), Q5
      return x0;
    } else {
      return sig[off + idx];
    }
  }

  // Samples idx, idx + 1 (idx even) of the even/odd packing.  ALIGNED: the pair sits on a 4-byte
  // boundary (even hop), one ds_read_b32; the sample before it is one more 2-byte read.
  template <bool PRE, bool ALIGNED>
  __device__ __forceinline__ cplx pair(int off, int idx) const {
    if constexpr (RAW16) {
      const lds_i16p s = (lds_i16p)(sigh + RAW_OFF + off + idx);
      float x0, x1;
      if constexpr (ALIGNED) {
        const int w = *(lds_i32p)s;
        x0 = (float)(short)w;
        x1 = (float)(w >> 16);
      } else {
        x0 = (float)s[0];
        x1 = (float)s[1];
      }
      if constexpr (PRE) return mk(x0 - cof * (float)s[-1], x1 - cof * x0);
      return mk(x0, x1);
    } else {
      return mk(sig[off + idx], sig[off + idx + 1]);
    }
  }

  template <int NSTEPS, bool PRE, bool ALIGNED>
  __device__ __forceinline__ void read(cplx (&v)[8]) const {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int first = SPLIT1024 ? 2 * (lane + 64 * a) : lane + 64 * a;
      if (a + 1 < NSTEPS) {
        if (SPLIT1024) {
          v[a] = pair<PRE, ALIGNED>(off_a, first);
        } else {
          const float yb = sample<PRE>(off_b, first);
          v[a] = mk(sample<PRE>(off_a, first), hasb ? yb : 0.f);
        }
      } else if (a + 1 == NSTEPS) {
        if (SPLIT1024) {
          const bool in0 = first < flen, in1 = first + 1 < flen;  // the sample after the frame is staged or slack: finite, masked
          const cplx y = pair<PRE, ALIGNED>(off_a, in0 ? first : 0);
          v[a] = mk(in0 ? y.x : 0.f, in1 ? y.y : 0.f);
        } else {
          const bool in = first < flen;
          const float xa = sample<PRE>(off_a, in ? first : 0), xb = sample<PRE>(off_b, in ? first : 0);
          v[a] = mk(in ? xa : 0.f, (in && hasb) ? xb : 0.f);
        }
      } else {
        v[a] = mk(0.f, 0.f);
      }
    }
  }

  template <bool PRE, bool ALIGNED>
  __device__ __forceinline__ void read_steps(int n_steps, cplx (&v)[8]) const {
    switch (n_steps) {  // wave-uniform; each case is straight-line code
      case 1: read<1, PRE, ALIGNED>(v); break;
      case 2: read<2, PRE, ALIGNED>(v); break;
      case 3: read<3, PRE, ALIGNED>(v); break;
      case 4: read<4, PRE, ALIGNED>(v); break;
      case 5: read<5, PRE, ALIGNED>(v); break;
      case 6: read<6, PRE, ALIGNED>(v); break;
      case 7: read<7, PRE, ALIGNED>(v); break;
      default: read<8, PRE, ALIGNED>(v); break;
    }
  }
};

// Natural log of a mel / frame energy.  With integer PCM an energy is either the eps that replaces
// an exact zero or at least ~1e-3 (one LSB), never a float32 denormal, so the bare v_log_f32
// (log2, 1 ulp) x ln 2 is safe: two instructions.  Float PCM can be arbitrarily quiet: keep the
// denormal-safe library expansion there.
template <bool INT_PCM>
__device__ __forceinline__ float fast_log(float x) {
  if constexpr (INT_PCM)
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
  else
    return __logf(x);
}

// Spec: configuration values fixed at compile time (>= 0) or read from the launch parameters (-1).
// The generic kernel (every field -1) serves any plan; the two standard front ends get an instance
// with their framing and output shape folded in, which removes the per-frame dispatch branches,
// most scalar bookkeeping and the scalar-register spills that came with it.
struct SpecGeneric {
  static constexpr int MAX_THREADS = 768;  // 12 waves: up to 168 VGPRs each
  static constexpr int STRIDE = -1, FLEN = -1, N_STEPS = -1, PRE = -1, KP = -1, NEED_ENERGY = -1, N_FT = -1, N_CT = -1,
                       OUT_KIND = -1, NFILT = -1, NCOLS = -1, DC_ELIM = -1;
};
// speechpy.feature.mfcc(fs 16 kHz, 20 ms / 10 ms, 40 filters, 13 cepstra, nfft 512), with or without
// the fused pre-emphasis (PRE stays a launch parameter: one wave-uniform branch per frame read)
struct SpecMfcc13 {
  static constexpr int MAX_THREADS = 1024;  // 16 waves of <= 128 VGPRs: this instance fits, and its LDS slice allows 15
  static constexpr int STRIDE = 160, FLEN = 320, N_STEPS = 5, PRE = -1, KP = 128, NEED_ENERGY = 1, N_FT = 3, N_CT = 1,
                       OUT_KIND = SVK_OUT_MFCC, NFILT = 40, NCOLS = 13, DC_ELIM = 1;
};
// the model's front end: lmfe(25 ms / 10 ms, 40 filters, nfft 1024) (load_data.py:64-70), with or without
// the fused pre-emphasis
struct SpecLmfe40 {
  static constexpr int MAX_THREADS = 768;
  static constexpr int STRIDE = 160, FLEN = 400, N_STEPS = 4, PRE = -1, KP = 256, NEED_ENERGY = 0, N_FT = 3, N_CT = 0,
                       OUT_KIND = SVK_OUT_LMFE, NFILT = 40, NCOLS = 40, DC_ELIM = 0;
};

template <typename PcmT, bool SPLIT1024, int TILE, bool RAW16, typename Spec>
__global__ __launch_bounds__(Spec::MAX_THREADS) void frontend_kernel(const FrontendParams p) {
  const int c_stride = Spec::STRIDE >= 0 ? Spec::STRIDE : p.stride;
  const int c_flen = Spec::FLEN >= 0 ? Spec::FLEN : p.flen;
  const int c_flen_eff = Spec::FLEN >= 0 ? Spec::FLEN : p.flen_eff;
  const int c_n_steps = Spec::N_STEPS >= 0 ? Spec::N_STEPS : p.n_steps;
  const int c_preemph = Spec::PRE >= 0 ? Spec::PRE : p.preemph;
  const int c_kp = Spec::KP >= 0 ? Spec::KP : p.kp;
  const int c_need_energy = Spec::NEED_ENERGY >= 0 ? Spec::NEED_ENERGY : p.need_energy;
  const int c_n_ft = Spec::N_FT >= 0 ? Spec::N_FT : p.n_ft;
  const int c_n_ct = Spec::N_CT >= 0 ? Spec::N_CT : p.n_ct;
  const int c_out_kind = Spec::OUT_KIND >= 0 ? Spec::OUT_KIND : p.out_kind;
  const int c_nfilt = Spec::NFILT >= 0 ? Spec::NFILT : p.nfilt;
  const int c_ncols = Spec::NCOLS >= 0 ? Spec::NCOLS : p.ncols;
  const int c_dc_elim = Spec::DC_ELIM >= 0 ? Spec::DC_ELIM : p.dc_elim;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // shared by the workgroup: mel and DCT operand fragments
  f32x4* fb_lds = reinterpret_cast<f32x4*>(smem);
  float* dct_lds = reinterpret_cast<float*>(smem + (size_t)p.n_slots * 64 * sizeof(f32x4));
  for (int i = threadIdx.x; i < p.n_slots * 64; i += blockDim.x) fb_lds[i] = p.fbfrag[i];
  for (int i = threadIdx.x; i < c_n_ct * c_n_ft * 4 * 64; i += blockDim.x) dct_lds[i] = p.dctfrag[i];
  int* tile_counter = reinterpret_cast<int*>(smem + p.table_bytes - 16);
  if (threadIdx.x == 0) *tile_counter = 0;
  __syncthreads();  // the only workgroup-wide barrier; from here on waves never wait for each other
  // private to this wave
  // the wave index is uniform: say so, or every per-tile index computation lands on the VALU
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
  char* mine = smem + p.table_bytes + (size_t)wave * p.wave_bytes;
  float* sig = reinterpret_cast<float*>(mine);
  int16_t* sigh = reinterpret_cast<int16_t*>(mine);
  float* ptile = reinterpret_cast<float*>(mine + p.sig_bytes);
  const int prow = c_kp + PT_PAD;
  // 8-frame tiles: the FFT scratch lies over rows 4..7 of the power tile (and a little beyond); the
  // spectra of frames 4..7 wait in registers (2-4 floats per lane and frame) until the tile's last FFT
  // is done.  nfft 1024 (wide rows, KP = 256): 4.2 KB less LDS per wave, 8 -> 12 waves per CU for the
  // model's front end; nfft 512: 2.2 KB less, 12 -> 15 waves for the MFCC instance.
  constexpr bool ALIAS_SCR = TILE == 8;
  cplx* scr = reinterpret_cast<cplx*>(ptile + (ALIAS_SCR ? 4 : TILE) * prow);
  const int lane_id = threadIdx.x & 63;
  constexpr bool INT_PCM = sizeof(PcmT) == 2;

  cplx t1[8], t2[8], t3[5];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    t1[r] = p.tw1[r * 64 + lane_id];
    t2[r] = p.tw2[r * 64 + lane_id];
  }
  if (SPLIT1024) {
#pragma unroll
    for (int q = 0; q < 5; ++q) t3[q] = p.tw3[q * 64 + lane_id];
  }
  // The workgroup owns tiles {r * grid * n_waves + block * n_waves + w}: n_waves consecutive tiles per
  // round r (neighbours share the overlapping PCM in L1).  Its waves take them in that order from a
  // counter in LDS rather than by a fixed stride: tiles differ in cost (a clip's last tile, clips
  // shortened by the VAD), and with fixed shares the waves of a CU finished up to 17 % apart.
  const int tpu = p.tiles_per_utt;
  const unsigned total_tiles = (unsigned)p.n_utt * (unsigned)tpu;  // < 2^31: checked by svk_frontend_run
  const unsigned round_stride = gridDim.x * (unsigned)n_waves;
  for (;;) {
    int ticket = 0;
    if (lane_id == 0) ticket = atomicAdd(tile_counter, 1);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    const unsigned round = (unsigned)ticket / (unsigned)n_waves;
    const unsigned tile = round * round_stride + blockIdx.x * (unsigned)n_waves + ((unsigned)ticket - round * (unsigned)n_waves);
    if (round > total_tiles / round_stride || tile >= total_tiles) break;  // the counter only grows: every wave gets here
    const int utt = (int)(tile / (unsigned)tpu), ft = (int)(tile - (unsigned)utt * (unsigned)tpu);
    // An opaque per-tile copy of the lane id: without it the compiler hoists every lane-derived
    // address and predicate of the tile body out of this loop and pins ~40 VGPRs for the whole
    // kernel (172 -> 3 waves per SIMD become 2); recomputing them per tile is a few dozen VALU.
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const int jf = lane & 15, g = lane >> 4;
    const int mirror = (64 - lane) & 63;
    const bool lane0 = lane == 0;
    const int f0 = ft * TILE;
    int64_t off = (int64_t)utt * p.clip_stride;  // utt is wave-uniform: these are scalar loads
    if (p.offsets) off = p.offsets[utt];
    int len = p.clip_len;
    if (p.lengths) len = p.lengths[utt];
    const int T = len >= c_flen ? (len - c_flen) / c_stride : 0;  // processing.py:115-116 (Q3)
    if (f0 == 0 && lane0 && p.n_frames) p.n_frames[utt] = T < p.max_frames ? T : p.max_frames;
    int nvalid = T - f0;
    nvalid = nvalid > TILE ? TILE : nvalid;
    if (f0 + nvalid > p.max_frames) nvalid = p.max_frames - f0;
    float* out_rows = p.feat + ((int64_t)utt * p.max_frames + f0) * c_ncols;

    if (nvalid > 0) {
      const PcmT* x = reinterpret_cast<const PcmT*>(p.pcm) + off;
      const int need = (nvalid - 1) * c_stride + c_flen_eff;
      wave_sync();  // previous tile's readers of sig / ptile are done
      if (SVK_ABLATE(p, 1)) {
      } else if constexpr (RAW16)
        stage_raw16(c_preemph != 0, reinterpret_cast<const int16_t*>(x), (int64_t)f0 * c_stride, need, len, sigh, lane,
                    p.src_chunk ? p.src_chunk + (int64_t)utt * p.chunk_stride : nullptr, p.chunk, p.chunk_inv);
      else
        stage_span<PcmT>(p, x, (int64_t)f0 * c_stride, need, len, sig, lane);
      wave_sync();

      // ---- spectra -----------------------------------------------------------
      // Two transforms per trip (4 frames at nfft 512, 2 at nfft 1024), pipelined through the one
      // scratch (fft512_wave_x2).  A missing second transform (ragged last tile of a clip) redoes
      // the first and drops the result: one code path.
      constexpr int FR_PER_FFT = SPLIT1024 ? 1 : 2;
      const bool pre = RAW16 && c_preemph != 0;
      auto load = [&](int f, cplx (&v)[8], float& ea, float& eb) __attribute__((always_inline)) {
        const bool hasb = !SPLIT1024 && f + 1 < nvalid;
        const FrameReader<SPLIT1024, RAW16> rd{sig, sigh, f * c_stride, (hasb ? f + 1 : f) * c_stride, c_flen_eff,
                                               p.pre_cof, hasb, lane};
        if (SPLIT1024 && RAW16 && ((f * c_stride) & 1) == 0) {  // RAW_OFF is even: the pairs are 4-byte aligned
          if (pre) rd.template read_steps<true, true>(c_n_steps, v);
          else rd.template read_steps<false, true>(c_n_steps, v);
        } else {
          if (pre) rd.template read_steps<true, false>(c_n_steps, v);
          else rd.template read_steps<false, false>(c_n_steps, v);
        }
        // Frame energy = sum over ALL nfft/2+1 power bins (feature.py:202).  By Parseval that is
        // sum(x^2)/2 + (X[0]^2 + X[nfft/2]^2) / (2 nfft), so only the bins the mel filters read
        // (k < kp) have to be untangled.
        // (skipped altogether when nobody reads the energy: log-mel output without d_energy)
        ea = 0.f, eb = 0.f;
        if (c_need_energy) {
          cplx e2 = mk(0.f, 0.f);  // (sum re^2, sum im^2): one packed fma per register
#pragma unroll
          for (int a = 0; a < 8; ++a) e2 = __builtin_elementwise_fma(v[a], v[a], e2);
          ea = SPLIT1024 ? 0.5f * (e2.x + e2.y) : 0.5f * e2.x;
          eb = 0.5f * e2.y;
        }
      };
      // defer = false_type: power bins (and energy group sums) go to the frame's tile row(s); true_type: into
      // hold[0..4] / hold[5..9] (first / second frame of the FFT, one float per 64-bin step) and hold[10], hold[11]
      // (their energy partials): see ALIAS_SCR.  Step j = 4 (bins 256..287) exists only when the bank reaches
      // bin 256 = nfft/4 of a 1024-point transform (kp = 288: SpeechPy's bank at any rate but 16 kHz, Q2).
      auto finish = [&](int f, cplx (&v)[8], float ea, float eb, float (&hold)[12], auto defer) __attribute__((always_inline)) {
        constexpr bool DEFER = decltype(defer)::value;
        if (lane0 && c_need_energy) {
          if (SPLIT1024) {  // X[0] = Re + Im, X[512] = Re - Im of Z[0]
            ea += (v[0].x * v[0].x + v[0].y * v[0].y) * (1.0f / 1024.0f);
          } else {          // X1[0], X1[256] = Re Z[0], Re Z[256];  X2: the imaginary parts
            ea += (v[0].x * v[0].x + v[4].x * v[4].x) * (1.0f / 1024.0f);
            eb += (v[0].y * v[0].y + v[4].y * v[4].y) * (1.0f / 1024.0f);
          }
        }
        float* rowa = ptile + f * prow;
        float* rowb = rowa + prow;  // (a lone last frame writes its zero partner into a row >= nvalid of the tile: masked at the output)
        cplx carry = v[0];  // lane 0 pairs bin 64 j with bin 64 (8 - j): one register later
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (64 * j < c_kp) {
            const cplx sm = shfl2(v[7 - j], mirror);
            const cplx zk = v[j], zn = lane0 ? carry : sm;
            carry = sm;
            const int k = lane + 64 * j;
            if (SPLIT1024) {
              // X[k] = E + W^k O  with  E = (Zk + conj Zn)/2,  O = (Zk - conj Zn)/(2i)
              // 2 E = Zk + conj Zn,  2 O = -i (Zk - conj Zn) = (zk.y + zn.y, -(zk.x - zn.x)); the 1/2's are folded into the filterbank weights
              const cplx E2 = add_conj(zk, zn), Ot = swap_add_conj(zk, zn);  // Ot = conj(2 O)
              cplx xp = E2 + cmul_conj(t3[j], Ot);
              xp *= xp;
              const float pk = xp.x + xp.y;  // 4 nfft |X[k]|^2
              if constexpr (DEFER) hold[j] = pk;
              else if (64 * (j + 1) <= c_kp) rowa[k] = pk;  // wave-uniform: no exec masking
              else if (k < c_kp) rowa[k] = pk;              // ragged last step only
            } else {
              // 2 X1 = Zk + conj Zn,  2i X2 = Zk - conj Zn  (|.|^2 is what matters)
              cplx xa = add_conj(zk, zn), xb = swap_add_conj(zk, zn);
              xa *= xa;
              xb *= xb;
              const float pa = xa.x + xa.y, pb = xb.x + xb.y;  // 4 nfft |X[k]|^2
              if constexpr (DEFER) {
                hold[j] = pa;
                hold[5 + j] = pb;
              } else if (64 * (j + 1) <= c_kp) {  // wave-uniform: no exec masking
                rowa[k] = pa;
                rowb[k] = pb;
              } else if (k < c_kp) {       // ragged last step only
                rowa[k] = pa;
                rowb[k] = pb;
              }
            }
          }
        }
        if (c_need_energy) {
          // 64 per-lane partial sums -> 8 (one per 8 lanes), parked in the 8 padding floats behind the
          // row's power bins; the mel stage adds them up with two more MFMAs against a matrix of ones.
          group8_sum2(ea, eb);  // (eb is idle for nfft 1024: its slot still hides ea's DPP wait states)
          if constexpr (DEFER) {
            hold[10] = ea;
            hold[11] = eb;
          } else if ((lane & 7) == 7) {
            rowa[c_kp + (lane >> 3)] = ea;
            if (!SPLIT1024) rowb[c_kp + (lane >> 3)] = eb;
          }
        }
      };
      if constexpr (ALIAS_SCR) {
        // two FFTs per trip: trips that fill rows 0..3 store directly, the later ones (rows 4..7, under the
        // scratch) keep their results in `held` -- unrolled so that `held` is indexed statically
        constexpr int TRIPS = 8 / (2 * FR_PER_FFT), DIRECT = TRIPS / 2, NHELD = 2 * (TRIPS - DIRECT);
        float held[NHELD][12] = {};
        float unused[12];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
          const int fa = 2 * FR_PER_FFT * t;
          if (fa < (SVK_ABLATE(p, 2) ? 0 : nvalid)) {  // wave-uniform
            const bool two = fa + FR_PER_FFT < nvalid;
            const int fb = two ? fa + FR_PER_FFT : fa;
            cplx va[8], vb[8];
            float eaa, eba, eab, ebb;
            load(fa, va, eaa, eba);
            load(fb, vb, eab, ebb);
            fft512_wave_x2<(Spec::N_STEPS > 0 ? Spec::N_STEPS : 8)>(va, vb, scr, lane, t1, t2);
            if (t < DIRECT) {
              finish(fa, va, eaa, eba, unused, std::false_type{});
              if (two) finish(fb, vb, eab, ebb, unused, std::false_type{});
            } else {
              finish(fa, va, eaa, eba, held[2 * (t - DIRECT)], std::true_type{});
              if (two) finish(fb, vb, eab, ebb, held[2 * (t - DIRECT) + 1], std::true_type{});
            }
          }
        }
        wave_sync();  // the scratch is dead: rows 4..7 may be written
#pragma unroll
        for (int q = 0; q < NHELD; ++q) {
          const int f = 4 + FR_PER_FFT * q;  // first frame of held FFT q
          if (f < (SVK_ABLATE(p, 2) ? 0 : nvalid)) {  // wave-uniform
            float* rowa = ptile + f * prow;
            float* rowb = rowa + prow;  // (nfft 512 only; a lone last frame's partner row is >= nvalid: masked at the output)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              const int k = lane + 64 * j;
              if (64 * (j + 1) <= c_kp || (64 * j < c_kp && k < c_kp)) {
                rowa[k] = held[q][j];
                if (!SPLIT1024) rowb[k] = held[q][5 + j];
              }
            }
            if (c_need_energy && (lane & 7) == 7) {
              rowa[c_kp + (lane >> 3)] = held[q][10];
              if (!SPLIT1024) rowb[c_kp + (lane >> 3)] = held[q][11];
            }
          }
        }
      } else {
        for (int fa = 0; fa < (SVK_ABLATE(p, 2) ? 0 : nvalid); fa += 2 * FR_PER_FFT) {
          const bool two = fa + FR_PER_FFT < nvalid;  // wave-uniform
          const int fb = two ? fa + FR_PER_FFT : fa;
          cplx va[8], vb[8];
          float eaa, eba, eab, ebb;
          load(fa, va, eaa, eba);
          load(fb, vb, eab, ebb);
          fft512_wave_x2<(Spec::N_STEPS > 0 ? Spec::N_STEPS : 8)>(va, vb, scr, lane, t1, t2);
          float unused[12];
          finish(fa, va, eaa, eba, unused, std::false_type{});
          if (two) finish(fb, vb, eab, ebb, unused, std::false_type{});
        }
      }
      wave_sync();

      if (SVK_ABLATE(p, 4)) continue;
      // ---- mel^T = fb x P^T (f32 MFMA), block-sparse over 16-bin chunks ---------
      f32x4 acc[MAX_FT];
      const float* pb = ptile + (jf & (TILE - 1)) * prow + 4 * g;  // an 8-frame tile repeats its rows in N = 8..15
#pragma unroll
      for (int t = 0; t < MAX_FT; ++t) {
        // two accumulators per filter tile: back-to-back MFMAs never wait on their own result
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if (t < c_n_ft && !SVK_ABLATE(p, 8)) {
          const f32x4* frag = fb_lds + (p.slot_base[t] - p.chunk_lo[t]) * 64 + lane;
          for (int u = p.chunk_lo[t]; u < p.chunk_hi[t]; u += 2) {  // the plan makes every chunk range even
            const f32x4 a0 = frag[u * 64], a1 = frag[(u + 1) * 64];
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(pb + 16 * u);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(pb + 16 * (u + 1));
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b0[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b0[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], b0[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], b0[3], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b1[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b1[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[2], b1[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[3], b1[3], acc1, 0, 0, 0);
          }
        }
        acc[t] = acc0 + acc1;
      }
      // frame energy = ones x (the 8 group sums of the frame): every output row of the tile is that sum,
      // so each lane ends up with the energy of ITS frame jf (feature.py:202-205)
      float e_frame = 0.f;
      if (c_need_energy) {
        const float* pe = ptile + (jf & (TILE - 1)) * prow + c_kp + g;
        f32x4 oe = (f32x4){0.f, 0.f, 0.f, 0.f};
        oe = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, pe[0], oe, 0, 0, 0);
        oe = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, pe[4], oe, 0, 0, 0);
        const float e = oe[0] * p.escale;
        e_frame = e == 0.f ? EPS64 : e;
      }
      if (SVK_ABLATE(p, 32)) continue;
      // lane (jf, g) now holds mel[filter 16 t + 4 g + reg][frame jf]
      const bool row_ok = jf < nvalid;
      float* orow = out_rows + (int64_t)jf * c_ncols;
#pragma unroll
      for (int t = 0; t < MAX_FT; ++t) {
        if (t < c_n_ft) {  // wave-uniform: an absent filter tile costs nothing
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float m = acc[t][r];
            m = m == 0.f ? EPS64 : m;  // feature.py:217
            if (c_out_kind != SVK_OUT_MFE) {
              m = fast_log<INT_PCM>(m);
              if (16 * t + 12 + 3 >= c_nfilt) m = (16 * t + 4 * g + r) < c_nfilt ? m : 0.f;  // only the ragged last tile
            }
            acc[t][r] = m;
          }
        }
      }
      if (c_out_kind != SVK_OUT_MFCC) {
#pragma unroll
        for (int t = 0; t < MAX_FT; ++t) {
          const int filt = 16 * t + 4 * g;
          if (row_ok && t < c_n_ft) {
            if ((c_ncols & 3) == 0 && filt + 3 < c_nfilt) {
              *reinterpret_cast<f32x4*>(orow + filt) = acc[t];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (filt + r < c_nfilt) orow[filt + r] = acc[t][r];
            }
          }
        }
      } else {
        // ---- cepstra^T = DCT x log(mel)^T: acc[t][r] is already the B operand ------
        const float le = fast_log<INT_PCM>(e_frame);
        for (int c = 0; c < (SVK_ABLATE(p, 16) ? 0 : c_n_ct); ++c) {  // runtime loop: keeps the table loads of one cepstral tile in flight, not four
          f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* dfrag = dct_lds + c * c_n_ft * 4 * 64 + lane;
#pragma unroll
          for (int t = 0; t < MAX_FT; ++t) {
            if (t < c_n_ft) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(dfrag[(t * 4 + r) * 64], acc[t][r], o, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cep = 16 * c + 4 * g + r;
            if (row_ok && cep < c_ncols) orow[cep] = (cep == 0 && c_dc_elim) ? le : o[r];  // feature.py:151-152
          }
        }
      }
      if (p.energy && row_ok && g == 0) p.energy[(int64_t)utt * p.max_frames + f0 + jf] = e_frame;
    }

    // rows of this tile past the clip's last frame: defined (zero) output
    const int first_bad = nvalid > 0 ? nvalid : 0;
    int last = p.max_frames - f0;
    last = last > TILE ? TILE : last;
    if (SVK_ABLATE(p, 64)) continue;
    for (int i = first_bad * c_ncols + lane; i < last * c_ncols; i += 64) out_rows[i] = 0.f;
    if (p.energy)
      for (int i = first_bad + lane; i < last; i += 64) p.energy[(int64_t)utt * p.max_frames + f0 + i] = 0.f;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------
// plan: device tables derived on the host, once per configuration
// ---------------------------------------------------------------------------------
struct svk_frontend_plan {
  svk_frontend_cfg cfg;
  int device;
  int kp, n_ft, n_ct, ncols, flen_eff;
  int tile;  // frames per wave tile (8 or 16), chosen for residency
  int n_slots, slot_base[MAX_FT];
  int table_bytes;
  int chunk_lo[MAX_FT], chunk_hi[MAX_FT];
  void* d_tables;  // one allocation: tw1 | tw2 | tw3 | fbfrag | dctfrag
  const cplx *tw1, *tw2, *tw3;
  const f32x4* fbfrag;
  const float* dctfrag;
};

namespace {

// LDS carve: [shared tables] then per wave [staged samples | power tile [tile][kp + PT_PAD] | FFT scratch]
struct LdsLayout {
  int sig_bytes, wave_bytes, waves;
  size_t total;
};
LdsLayout lds_layout(const svk_frontend_plan* plan, int tile, bool raw16, int lds_per_cu, int max_waves = 12) {
  const bool alias_scr = tile == 8;  // frontend_kernel: ALIAS_SCR
  const int span = (tile - 1) * plan->cfg.frame_stride + plan->flen_eff;
  LdsLayout l;
  if (raw16)
    l.sig_bytes = (((RAW_OFF + span + 8) * 2 + 15) / 16) * 16;  // whole 16-byte groups per lane
  else
    l.sig_bytes = (((span + 8) * 4 + 15) / 16) * 16;
  const int prow = plan->kp + PT_PAD;
  if (alias_scr)  // scratch starts at row 4 of the power tile
    l.wave_bytes = l.sig_bytes + (int)sizeof(float) * std::max(tile * prow, 4 * prow + 2 * SCR);
  else
    l.wave_bytes = l.sig_bytes + (int)sizeof(float) * (tile * prow + 2 * SCR);
  l.wave_bytes = ((l.wave_bytes + 15) / 16) * 16;
  const int room = lds_per_cu - plan->table_bytes;
  // 12 waves = 768 threads: up to 168 VGPRs each; an instance built for 1 024 threads (<= 128 VGPRs) may take 16
  l.waves = room >= l.wave_bytes ? std::min(max_waves, room / l.wave_bytes) : 0;
  l.total = (size_t)plan->table_bytes + (size_t)l.waves * l.wave_bytes;
  return l;
}

template <typename PcmT, bool RAW16>
void (*pick_kernel(bool split, int tile))(const FrontendParams) {
  if (split)
    return tile == 8 ? frontend_kernel<PcmT, true, 8, RAW16, SpecGeneric> : frontend_kernel<PcmT, true, 16, RAW16, SpecGeneric>;
  return tile == 8 ? frontend_kernel<PcmT, false, 8, RAW16, SpecGeneric> : frontend_kernel<PcmT, false, 16, RAW16, SpecGeneric>;
}

// Does the launch match every value a specialised instance has folded in?
template <typename Spec>
bool spec_matches(const FrontendParams& p) {
  return p.stride == Spec::STRIDE && p.flen == Spec::FLEN && p.flen_eff == Spec::FLEN && p.n_steps == Spec::N_STEPS &&
         (Spec::PRE < 0 || (p.preemph != 0) == (Spec::PRE != 0)) && (!p.preemph || p.pre_shift == 1) && p.kp == Spec::KP &&
         (p.need_energy != 0) == (Spec::NEED_ENERGY != 0) && p.n_ft == Spec::N_FT && p.n_ct == Spec::N_CT &&
         p.out_kind == Spec::OUT_KIND && p.nfilt == Spec::NFILT && p.ncols == Spec::NCOLS &&
         (Spec::OUT_KIND != SVK_OUT_MFCC || (p.dc_elim != 0) == (Spec::DC_ELIM != 0));
}

}  // namespace

extern "C" {

int64_t svk_frontend_num_frames(const svk_frontend_cfg* cfg, int64_t n_samples) {
  if (!cfg || cfg->frame_stride <= 0 || n_samples < cfg->frame_len) return 0;
  return (n_samples - cfg->frame_len) / cfg->frame_stride;
}

int svk_frontend_num_cols(const svk_frontend_cfg* cfg) {
  if (!cfg) return 0;
  return cfg->out_kind == SVK_OUT_MFCC ? cfg->num_ceps : cfg->num_filters;
}

int svk_frontend_plan_create(svk_ctx* ctx, const svk_frontend_cfg* cfg, const double* h_filterbank,
                             svk_frontend_plan** out) {
  if (!ctx || !cfg || !h_filterbank || !out) return SVK_ERR_BAD_ARG;
  *out = nullptr;
  SVK_REQUIRE(ctx, cfg->frame_len >= 1 && cfg->frame_stride >= 1, "frame_len / frame_stride must be >= 1");
  SVK_REQUIRE(ctx, cfg->num_filters >= 1, "num_filters must be >= 1");
  SVK_REQUIRE(ctx, cfg->out_kind >= SVK_OUT_MFE && cfg->out_kind <= SVK_OUT_MFCC, "out_kind");
  if (cfg->nfft != 512 && cfg->nfft != 1024)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "fused front end supports fft_length 512 or 1024, got %d", cfg->nfft);
  if (cfg->num_filters > 16 * MAX_FT)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "fused front end supports up to %d filters, got %d", 16 * MAX_FT,
                    cfg->num_filters);
  if (cfg->out_kind == SVK_OUT_MFCC)
    SVK_REQUIRE(ctx, cfg->num_ceps >= 1 && cfg->num_ceps <= cfg->num_filters, "1 <= num_ceps <= num_filters");

  svk_frontend_plan* plan = new (std::nothrow) svk_frontend_plan();
  if (!plan) return SVK_ERR_OOM;
  plan->cfg = *cfg;
  plan->device = ctx->device;
  const int nbins = cfg->nfft / 2 + 1;
  const int nf = cfg->num_filters;
  plan->ncols = svk_frontend_num_cols(cfg);
  plan->n_ft = (nf + 15) / 16;
  plan->n_ct = cfg->out_kind == SVK_OUT_MFCC ? (cfg->num_ceps + 15) / 16 : 0;
  plan->flen_eff = cfg->frame_len < cfg->nfft ? cfg->frame_len : cfg->nfft;

  // highest bin with a non-zero weight decides how many power bins are kept
  int kmax = -1;
  for (int i = 0; i < nf; ++i)
    for (int k = 0; k < nbins; ++k)
      if (h_filterbank[(size_t)i * nbins + k] != 0.0 && k > kmax) kmax = k;
  // bins 0..255 are the four 64-bin steps of the untangling; bin 256 (= nfft/4 of a 1024-point transform, where
  // SpeechPy's bank ends at every sampling rate but 16 kHz, Q2) costs a fifth, mostly idle step: kp = 288
  if (kmax > 256) {
    delete plan;
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED,
                    "filterbank reaches bin %d; the fused kernel keeps bins <= 256 (the SpeechPy bank stops at "
                    "(nfft/2+2)/2, Q2)", kmax);
  }
  plan->kp = ((kmax + 1 + 31) / 32) * 32;  // a multiple of 32 bins: the mel loop consumes chunks of 16 in pairs
  if (plan->kp < 32) plan->kp = 32;
  const int nchunks = plan->kp / 16;
  for (int t = 0; t < MAX_FT; ++t) {
    int lo = nchunks, hi = 0;
    for (int i = 16 * t; i < 16 * t + 16 && i < nf; ++i)
      for (int k = 0; k < plan->kp && k < nbins; ++k)
        if (h_filterbank[(size_t)i * nbins + k] != 0.0) {
          lo = std::min(lo, k / 16);
          hi = std::max(hi, k / 16 + 1);
        }
    if (hi <= lo) lo = hi = 0;
    if ((hi - lo) & 1) {  // pair up: take one more (all-zero) chunk on whichever side has room
      if (hi < nchunks) ++hi; else --lo;
    }
    plan->chunk_lo[t] = lo;
    plan->chunk_hi[t] = hi;
  }
  plan->n_slots = 0;
  for (int t = 0; t < MAX_FT; ++t) {
    plan->slot_base[t] = plan->n_slots;
    plan->n_slots += plan->chunk_hi[t] - plan->chunk_lo[t];
  }
  plan->table_bytes = plan->n_slots * 64 * (int)sizeof(f32x4) +
                      std::max(plan->n_ct, 0) * plan->n_ft * 4 * 64 * (int)sizeof(float);
  plan->table_bytes = ((plan->table_bytes + 15) / 16) * 16 + 16;  // + the workgroup's tile counter
  // Tile size: 8 frames leave half of the MFMA N dimension idle but halve the per-wave LDS slice,
  // i.e. double the waves a CU can hold.  SVK_FRONTEND_TILE=8|16 overrides (tuning).
  {
    const int w16 = lds_layout(plan, 16, true, ctx->lds_per_cu).waves, w8 = lds_layout(plan, 8, true, ctx->lds_per_cu).waves;
    plan->tile = w8 > w16 ? 8 : 16;
    if (const char* env = getenv("SVK_FRONTEND_TILE")) {
      const int t = atoi(env);
      if (t == 8 || t == 16) plan->tile = t;
    }
  }
  if (lds_layout(plan, plan->tile, false, ctx->lds_per_cu).waves < 1) {
    delete plan;
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "frame stride %d / %d filters need more than %d bytes of LDS per CU",
                    cfg->frame_stride, cfg->num_filters, ctx->lds_per_cu);
  }

  // ---- host tables (float64 maths, rounded once to f32) ----
  const size_t n_tw = 8 * 64, n_tw3 = 5 * 64;
  const size_t n_fb = (size_t)std::max(plan->n_slots, 1) * 64 * 4;
  const size_t n_dct = (size_t)std::max(plan->n_ct, 1) * plan->n_ft * 4 * 64;
  std::vector<float> host(2 * n_tw * 2 + n_tw3 * 2 + n_fb + n_dct, 0.f);
  float* h_tw1 = host.data();
  float* h_tw2 = h_tw1 + 2 * n_tw;
  float* h_tw3 = h_tw2 + 2 * n_tw;
  float* h_fb = h_tw3 + 2 * n_tw3;
  float* h_dct = h_fb + n_fb;
  const double PI = 3.14159265358979323846;
  for (int r = 0; r < 8; ++r)
    for (int l = 0; l < 64; ++l) {
      double a1 = -2.0 * PI * (double)(l * r) / 512.0;
      double a2 = -2.0 * PI * (double)((l & 7) * r) / 64.0;
      h_tw1[2 * (r * 64 + l)] = (float)cos(a1);
      h_tw1[2 * (r * 64 + l) + 1] = (float)sin(a1);
      h_tw2[2 * (r * 64 + l)] = (float)cos(a2);
      h_tw2[2 * (r * 64 + l) + 1] = (float)sin(a2);
    }
  for (int q = 0; q < 5; ++q)
    for (int l = 0; l < 64; ++l) {
      double a = -2.0 * PI * (double)(l + 64 * q) / 1024.0;
      h_tw3[2 * (q * 64 + l)] = (float)cos(a);
      h_tw3[2 * (q * 64 + l) + 1] = (float)sin(a);
    }
  // input_scale (e.g. 2^-15: int16 PCM read the way librosa hands it to the reference's lmfe call,
  // load_data.py:50-70) enters every power as its square: folded into the weights, exact for powers of two
  const double in_scale = cfg->input_scale != 0.f ? (double)cfg->input_scale : 1.0;
  const double power_scale = (cfg->nfft == 1024 ? 1.0 / 4096.0 : 1.0 / 2048.0) * in_scale * in_scale;
  // A-operand fragments of the filterbank: lane l = (i = l & 15, kk = l >> 4), element e of
  // chunk u is fb[16 t + i][16 u + 4 kk + e]
  for (int t = 0; t < plan->n_ft; ++t)
    for (int u = plan->chunk_lo[t]; u < plan->chunk_hi[t]; ++u)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
          const int filt = 16 * t + (l & 15), bin = 16 * u + 4 * (l >> 4) + e;
          double w = (filt < nf && bin < nbins) ? h_filterbank[(size_t)filt * nbins + bin] : 0.0;
          // the weights carry the spectrum's 1/nfft and the 1/4 of the untangling (powers of two: exact),
          // so the kernel stores raw |.|^2 sums in the power tile
          h_fb[(((size_t)plan->slot_base[t] + (u - plan->chunk_lo[t])) * 64 + l) * 4 + e] = (float)(w * power_scale);
        }
  // A-operand fragments of the DCT-II (ortho) matrix, k-step (t, r) <-> filter 16 t + 4 (l >> 4) + r
  // scipy.fftpack.dct(type=2, norm='ortho'): D[k][n] = sqrt(2/N) cos(pi k (2n+1) / 2N), D[0][n] = sqrt(1/N)
  for (int c = 0; c < plan->n_ct; ++c)
    for (int t = 0; t < plan->n_ft; ++t)
      for (int r = 0; r < 4; ++r)
        for (int l = 0; l < 64; ++l) {
          const int cep = 16 * c + (l & 15), filt = 16 * t + 4 * (l >> 4) + r;
          double d = 0.0;
          if (cep < cfg->num_ceps && filt < nf)
            d = cep == 0 ? sqrt(1.0 / nf) : sqrt(2.0 / nf) * cos(PI * cep * (2.0 * filt + 1.0) / (2.0 * nf));
          h_dct[(((size_t)c * plan->n_ft + t) * 4 + r) * 64 + l] = (float)d;
        }

  if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&plan->d_tables, host.size() * sizeof(float)) != hipSuccess) {
    delete plan;
    return svk_fail(ctx, SVK_ERR_HIP, "hipMalloc of front-end tables failed");
  }
  if (hipMemcpy(plan->d_tables, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(plan->d_tables);
    delete plan;
    return svk_fail(ctx, SVK_ERR_HIP, "upload of front-end tables failed");
  }
  float* d = reinterpret_cast<float*>(plan->d_tables);
  plan->tw1 = reinterpret_cast<const cplx*>(d);
  plan->tw2 = reinterpret_cast<const cplx*>(d + 2 * n_tw);
  plan->tw3 = reinterpret_cast<const cplx*>(d + 4 * n_tw);
  plan->fbfrag = reinterpret_cast<const f32x4*>(d + 4 * n_tw + 2 * n_tw3);
  plan->dctfrag = d + 4 * n_tw + 2 * n_tw3 + n_fb;
  *out = plan;
  return SVK_OK;
}

void svk_frontend_plan_destroy(svk_frontend_plan* plan) {
  if (!plan) return;
  if (plan->d_tables) (void)hipFree(plan->d_tables);
  delete plan;
}

int svk_frontend_run(svk_ctx* ctx, const svk_frontend_plan* plan, const void* d_pcm, int pcm_dtype,
                     const int64_t* d_offsets, const int32_t* d_lengths, int64_t clip_stride, int32_t clip_len,
                     int32_t n_utt, int32_t max_frames, float* d_feat, float* d_energy, int32_t* d_n_frames,
                     const int32_t* d_src_chunk, int32_t chunk_samples, int32_t chunk_stride) {
  if (!ctx || !plan) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, plan->device == ctx->device, "plan belongs to another device");
  SVK_REQUIRE(ctx, n_utt >= 0 && max_frames >= 0, "n_utt / max_frames negative");
  if (n_utt == 0 || max_frames == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_pcm && d_feat, "d_pcm / d_feat is NULL");
  SVK_REQUIRE(ctx, pcm_dtype == SVK_PCM_I16 || pcm_dtype == SVK_PCM_F32, "pcm_dtype");
  SVK_REQUIRE(ctx, d_offsets || clip_stride >= 0, "clip_stride negative");
  SVK_REQUIRE(ctx, d_lengths || clip_len >= 0, "clip_len negative");

  FrontendParams p;
  p.pcm = d_pcm;
  p.offsets = d_offsets;
  p.lengths = d_lengths;
  p.clip_stride = clip_stride;
  p.clip_len = clip_len;
  p.n_utt = n_utt;
  p.max_frames = max_frames;
  const bool raw16 = pcm_dtype == SVK_PCM_I16 && (!plan->cfg.preemph || plan->cfg.preemph_shift == 1) &&
                     !(getenv("SVK_FE_F32STAGE") && atoi(getenv("SVK_FE_F32STAGE")));
  p.src_chunk = d_src_chunk;
  p.chunk = chunk_samples;
  p.chunk_stride = chunk_stride;
  p.chunk_inv = chunk_samples > 0 ? 1.0f / (float)chunk_samples : 0.f;
  if (d_src_chunk) {
    SVK_REQUIRE(ctx, (int64_t)chunk_samples * chunk_stride < ((int64_t)1 << 24), "gathered clips of at most 2^24 samples");
    SVK_REQUIRE(ctx, d_lengths, "gathered input needs d_lengths (the gathered length of every clip)");
    if (!raw16 || chunk_samples < 8 || (chunk_samples & 7) || chunk_stride < 0)
      return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "gathered input (d_src_chunk) is built for int16 PCM through the 16-bit staging path "
                      "and chunks of a multiple of 8 samples (got dtype %d, chunk %d)", pcm_dtype, chunk_samples);
  }
  const LdsLayout lds = lds_layout(plan, plan->tile, raw16, ctx->lds_per_cu);
  p.tiles_per_utt = (max_frames + plan->tile - 1) / plan->tile;
  p.flen = plan->cfg.frame_len;
  p.flen_eff = plan->flen_eff;
  p.stride = plan->cfg.frame_stride;
  p.nfilt = plan->cfg.num_filters;
  p.ncols = plan->ncols;
  p.out_kind = plan->cfg.out_kind;
  p.dc_elim = plan->cfg.dc_elimination;
  p.preemph = plan->cfg.preemph;
  p.pre_shift = plan->cfg.preemph_shift;
  p.pre_cof = plan->cfg.preemph_cof;
  p.escale = plan->cfg.input_scale != 0.f ? plan->cfg.input_scale * plan->cfg.input_scale : 1.0f;
  p.kp = plan->kp;
  p.sig_bytes = lds.sig_bytes;
  p.wave_bytes = lds.wave_bytes;
  p.table_bytes = plan->table_bytes;
  {
    const int per = plan->cfg.nfft == 1024 ? 128 : 64;
    p.n_steps = std::max(1, std::min(8, (plan->flen_eff + per - 1) / per));
  }
  p.ablate = getenv("SVK_FE_ABLATE") ? atoi(getenv("SVK_FE_ABLATE")) : 0;
  p.n_slots = plan->n_slots;
  for (int t = 0; t < MAX_FT; ++t) p.slot_base[t] = plan->slot_base[t];
  p.n_ft = plan->n_ft;
  p.n_ct = plan->n_ct;
  for (int t = 0; t < MAX_FT; ++t) {
    p.chunk_lo[t] = plan->chunk_lo[t];
    p.chunk_hi[t] = plan->chunk_hi[t];
  }
  p.tw1 = plan->tw1;
  p.tw2 = plan->tw2;
  p.tw3 = plan->tw3;
  p.fbfrag = plan->fbfrag;
  p.dctfrag = plan->dctfrag;
  p.feat = d_feat;
  p.energy = d_energy;
  p.n_frames = d_n_frames;
  p.need_energy = (d_energy != nullptr) || (plan->cfg.out_kind == SVK_OUT_MFCC && plan->cfg.dc_elimination);

  const int64_t total = (int64_t)n_utt * p.tiles_per_utt;
  const bool generic_only = getenv("SVK_FE_GENERIC") != nullptr;  // tuning / tests: force the unspecialised instance
  const bool split = plan->cfg.nfft == 1024;
  // the int16 MFCC instance is compiled for 1 024 threads (SpecMfcc13::MAX_THREADS)
  const bool mfcc13_i16 = pcm_dtype == SVK_PCM_I16 && raw16 && plan->tile == 8 && !generic_only && !split &&
                          spec_matches<SpecMfcc13>(p);
  const int wave_cap = mfcc13_i16 ? SpecMfcc13::MAX_THREADS / 64 : 12;
  const int fit_waves = mfcc13_i16 ? lds_layout(plan, plan->tile, raw16, ctx->lds_per_cu, wave_cap).waves : lds.waves;
  // one workgroup of that many waves per CU (it owns the CU's LDS); fewer when there is little work
  int waves = (int)std::max<int64_t>(1, std::min<int64_t>(fit_waves, (total + ctx->num_cu - 1) / ctx->num_cu));
  if (const char* env = getenv("SVK_FE_WAVES")) waves = std::max(1, std::min(waves, atoi(env)));  // tuning only
  const int64_t grid = std::min<int64_t>((total + waves - 1) / waves, ctx->num_cu);
  if ((int64_t)n_utt * p.tiles_per_utt >= ((int64_t)1 << 31))
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "%lld frame tiles in one launch (limit 2^31): split the batch",
                    (long long)n_utt * p.tiles_per_utt);
  const size_t lds_total = (size_t)plan->table_bytes + (size_t)waves * lds.wave_bytes;
  void (*kern)(const FrontendParams) = nullptr;
  if (mfcc13_i16)
    kern = frontend_kernel<int16_t, false, 8, true, SpecMfcc13>;
  else if (pcm_dtype == SVK_PCM_I16 && raw16 && plan->tile == 8 && !generic_only && split && spec_matches<SpecLmfe40>(p))
    kern = frontend_kernel<int16_t, true, 8, true, SpecLmfe40>;
  else if (pcm_dtype == SVK_PCM_I16)
    kern = raw16 ? pick_kernel<int16_t, true>(split, plan->tile) : pick_kernel<int16_t, false>(split, plan->tile);
  // float32 signals (what librosa hands the reference's lmfe call, load_data.py:50-70): the same two configurations
  else if (plan->tile == 8 && !generic_only && !split && spec_matches<SpecMfcc13>(p))
    kern = frontend_kernel<float, false, 8, false, SpecMfcc13>;
  else if (plan->tile == 8 && !generic_only && split && spec_matches<SpecLmfe40>(p))
    kern = frontend_kernel<float, true, 8, false, SpecLmfe40>;
  else
    kern = pick_kernel<float, false>(split, plan->tile);
  if (getenv("SVK_FE_DEBUG"))
    fprintf(stderr, "svk_frontend_run: %s instance, %d waves/CU, grid %lld\n",
            kern == (void (*)(const FrontendParams))frontend_kernel<int16_t, false, 8, true, SpecMfcc13>   ? "mfcc13"
            : kern == (void (*)(const FrontendParams))frontend_kernel<int16_t, true, 8, true, SpecLmfe40> ? "lmfe40"
            : kern == (void (*)(const FrontendParams))frontend_kernel<float, false, 8, false, SpecMfcc13> ? "mfcc13 (f32)"
            : kern == (void (*)(const FrontendParams))frontend_kernel<float, true, 8, false, SpecLmfe40>  ? "lmfe40 (f32)"
                                                                                                           : "generic",
            waves, (long long)grid);
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_total));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), lds_total, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

}  // extern "C"
