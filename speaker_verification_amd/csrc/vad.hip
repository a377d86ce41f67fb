// Energy VAD + hysteresis + compaction, one workgroup per clip.
//
// Replaces /root/reference/vad.py:44-57 (30 ms framer; `while offset + n < len`, Q12) and
// vad.py:60-129 (ring-buffer trigger / release, Q13).  The per-frame decision the reference
// delegates to webrtcvad (vad.py:90) is this build's integer rule
//     sum(x^2) > threshold * frame_samples                       (int64, bit-exact)
// Phase 1 (all waves): per-frame sum of squares with 16-byte loads and a wave reduction.
// Phase 2 (wave 0, wave-uniform): the hysteresis is inherently sequential over <= a few thousand
//     frames; the flags travel as 64-bit ballot masks and the walk is scalar arithmetic; it
//     records keep / segment / packed position per frame.
// Phase 3 (all waves): kept frames are copied to the front of the clip's output slot.
// HBM-bound: 2 bytes read per sample (+2 written when compacting).
#include <cstdlib>

#include "svk_internal.h"

namespace {

constexpr int MAX_VAD_FRAMES = 8192;  // LDS budget: 245 s of 30 ms frames per clip
constexpr int VAD_THREADS = 256;

typedef short i16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(VAD_THREADS) void vad_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ offsets,
                                                  const int32_t* __restrict__ lengths, int64_t clip_stride,
                                                  int clip_len, int fsamp, int ring_len, int ring_thresh,
                                                  long long threshold, int max_vf, uint8_t* __restrict__ keep_out,
                                                  int32_t* __restrict__ seg_out, int32_t* __restrict__ nvf_out,
                                                  int16_t* __restrict__ voiced, int32_t* __restrict__ voiced_len, int32_t* __restrict__ src_frame) {
  __shared__ uint8_t flag[MAX_VAD_FRAMES];
  __shared__ int32_t pos[MAX_VAD_FRAMES];  // packed frame index of a kept frame, or -1
  const int utt = blockIdx.x;
  const int64_t off = offsets ? offsets[utt] : (int64_t)utt * clip_stride;
  const int len = lengths ? lengths[utt] : clip_len;
  // frames yielded while offset + n < len(audio) in BYTES: n = 2 fsamp, len = 2 L  (vad.py:54)
  int nf = len > 0 ? (int)((2LL * len - 1) / (2LL * fsamp)) : 0;
  if (nf > max_vf) nf = max_vf;
  const int16_t* x = pcm + off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = VAD_THREADS / 64;

  // Usual geometry (30 ms at 16 kHz = 480 samples = 60 16-byte vectors, clip on a 16-byte boundary):
  // a frame is ONE vector per lane, and a wave keeps eight frames' loads in flight (one load per
  // trip followed by its own reduction paid a full memory round trip per frame).
  const int nvec_f = fsamp >> 3;
  const bool one_vec = (fsamp & 7) == 0 && nvec_f <= 64 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  constexpr int VU = 8;
  for (int f0 = wave; one_vec && f0 < nf; f0 += NW * VU) {
    i16x8 v[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int f = f0 + NW * u;
      const bool on = f < nf && lane < nvec_f;
      v[u] = *reinterpret_cast<const i16x8*>(x + (on ? (int64_t)f * fsamp + 8 * lane : 0));
      if (!on) v[u] = (i16x8)(short)0;
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int f = f0 + NW * u;
      if (f >= nf) break;  // wave-uniform
      long long acc = 0;   // squares of 16-bit values pair up in 32 bits: a^2 + b^2 <= 2^31
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        acc += (long long)((unsigned)((int)v[u][e] * (int)v[u][e]) + (unsigned)((int)v[u][e + 1] * (int)v[u][e + 1]));
      acc = wave_sum(acc);
      if (lane == 0) flag[f] = acc > threshold * (long long)fsamp ? 1 : 0;
    }
  }
  for (int f = wave; !one_vec && f < nf; f += NW) {
    const int16_t* fr = x + (int64_t)f * fsamp;
    long long acc = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(fr) & 15) == 0;
    if (aligned) {
      const int nvec = fsamp >> 3;
      for (int i = lane; i < nvec; i += 64) {
        const i16x8 v = *reinterpret_cast<const i16x8*>(fr + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (long long)((int)v[e] * (int)v[e]);
      }
      for (int i = (nvec << 3) + lane; i < fsamp; i += 64) acc += (long long)((int)fr[i] * (int)fr[i]);
    } else {
      for (int i = lane; i < fsamp; i += 64) acc += (long long)((int)fr[i] * (int)fr[i]);
    }
    acc = wave_sum(acc);
    if (lane == 0) flag[f] = acc > threshold * (long long)fsamp ? 1 : 0;
  }
  __syncthreads();

  if (wave == 0) {
    // The hysteresis is sequential, but nothing in it is per-lane: the whole of wave 0 walks it with
    // wave-uniform (scalar) state.  64 frame flags at a time become one ballot mask; lane b keeps
    // the result of frame f0 + b in a register and the chunk is written back with one store per
    // lane.  (A single lane reading flag[f] and flag[f - ring_len] out of LDS every step was one
    // LDS round trip per frame.)
    // ring buffer = the last min(since_clear, ring_len) frames since it was last cleared
    bool triggered = false;
    int since = 0, voiced_in_ring = 0, kept = 0, seg = 0;
    unsigned long long prev = 0;
    for (int f0 = 0; f0 < nf; f0 += 64) {
      const int fl = f0 + lane;
      const unsigned long long m = __ballot(fl < nf && flag[fl] != 0);
      int my_pos = -1, my_seg = -1;
      const int cnt = nf - f0 < 64 ? nf - f0 : 64;
      for (int b = 0; b < cnt; ++b) {
        const int f = f0 + b;
        const int sbit = (int)((m >> b) & 1ull);
        if (since == ring_len) {  // deque(maxlen) drops the oldest: flag[f - ring_len]
          // this chunk's or the previous chunk's ballot mask; a ring longer than 64 frames (10 ms
          // frames with 1 s of padding: 100) reaches further back: one LDS broadcast read
          const int o = b - ring_len;
          voiced_in_ring -= o >= 0     ? (int)((m >> o) & 1ull)
                            : o >= -64 ? (int)((prev >> (64 + o)) & 1ull)
                                       : (int)flag[f - ring_len];
        } else {
          ++since;
        }
        voiced_in_ring += sbit;
        int v_pos = -1, v_seg = -1;
        if (!triggered) {
          if (voiced_in_ring > ring_thresh) {  // vad.py:99  num_voiced > 0.9 * maxlen
            triggered = true;
            for (int g = f - since + 1; g <= f; ++g) {  // vad.py:105-106: the whole ring is emitted
              if (lane == (g & 63)) {                   // the lane that owns frame g (f0 is a multiple of 64)
                if (g >= f0) {
                  my_pos = kept;
                  my_seg = seg;
                } else {  // frame of the previous chunk: already written back as -1
                  pos[g] = kept;
                  if (seg_out) seg_out[(int64_t)utt * max_vf + g] = seg;
                }
              }
              ++kept;
            }
            since = 0;
            voiced_in_ring = 0;
            continue;
          }
        } else {
          v_seg = seg;
          v_pos = kept++;
          if (since - voiced_in_ring > ring_thresh) {  // vad.py:117  num_unvoiced > 0.9 * maxlen
            triggered = false;
            ++seg;
            since = 0;
            voiced_in_ring = 0;
          }
        }
        if (lane == b) {
          my_pos = v_pos;
          my_seg = v_seg;
        }
      }
      if (fl < nf) {
        pos[fl] = my_pos;
        if (seg_out) seg_out[(int64_t)utt * max_vf + fl] = my_seg;
      }
      prev = m;
    }
    if (lane == 0) {
      if (nvf_out) nvf_out[utt] = nf;
      if (voiced_len) voiced_len[utt] = kept * fsamp;
    }
  }
  __syncthreads();

  for (int f = threadIdx.x; f < max_vf; f += VAD_THREADS) {
    keep_out[(int64_t)utt * max_vf + f] = (f < nf && pos[f] >= 0) ? 1 : 0;
    if (seg_out && f >= nf) seg_out[(int64_t)utt * max_vf + f] = -1;
    // the index form of the compaction: src_frame[q] = the frame that is the q-th kept one (svk_frontend_run reads through it)
    if (src_frame && f < nf && pos[f] >= 0) src_frame[(int64_t)utt * max_vf + pos[f]] = f;
  }
  if (voiced) {
    int16_t* dst = voiced + off;
    const bool one_vec_copy = one_vec && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int f0 = wave; one_vec_copy && f0 < nf; f0 += NW * VU) {  // eight frames' loads first, then their stores
      i16x8 v[VU];
      int pf[VU];
#pragma unroll
      for (int u = 0; u < VU; ++u) {
        const int f = f0 + NW * u;
        pf[u] = f < nf ? pos[f] : -1;
        if (pf[u] >= 0 && lane < nvec_f) v[u] = *reinterpret_cast<const i16x8*>(x + (int64_t)f * fsamp + 8 * lane);
      }
#pragma unroll
      for (int u = 0; u < VU; ++u)
        if (pf[u] >= 0 && lane < nvec_f) *reinterpret_cast<i16x8*>(dst + (int64_t)pf[u] * fsamp + 8 * lane) = v[u];
    }
    for (int f = wave; !one_vec_copy && f < nf; f += NW) {
      const int pf = pos[f];
      if (pf < 0) continue;
      const int16_t* s = x + (int64_t)f * fsamp;
      int16_t* d = dst + (int64_t)pf * fsamp;
      const bool aligned = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
      if (aligned) {
        const int nvec = fsamp >> 3;
        for (int i = lane; i < nvec; i += 64)
          *reinterpret_cast<i16x8*>(d + 8 * i) = *reinterpret_cast<const i16x8*>(s + 8 * i);
        for (int i = (nvec << 3) + lane; i < fsamp; i += 64) d[i] = s[i];
      } else {
        for (int i = lane; i < fsamp; i += 64) d[i] = s[i];
      }
    }
  }
}

// ---- LONG clips (VoxCeleb utterances run to 145 s = 4 833 frames): one workgroup per clip leaves a batch of a few long
// clips to a few workgroups that each stream megabytes and then walk thousands of frames (measured: 1.57 ms for 14 clips
// of 31 .. 145 s).  The same three phases as three kernels: the frame energies and the compaction copy are cut into
// chunks of VAD_CHUNK frames over the whole grid; only the hysteresis stays one wave per clip (it is sequential by
// definition; ~30 scalar instructions per frame).  Flags and packed positions travel through the handle's workspace.
// Usual geometry only (frame = one 16-byte vector per lane); anything else keeps the one-kernel path.  Bit-exact: the
// same integer rule, the same walk. ----
constexpr int VAD_CHUNK = 32;   // frames per workgroup: 4 waves x 8 frames in flight

__global__ __launch_bounds__(VAD_THREADS) void vad_flags_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ lengths, int64_t clip_stride, int clip_len,
                                                                int fsamp, long long threshold, int max_vf, uint8_t* __restrict__ flag_g) {
  const int utt = blockIdx.y;
  const int64_t off = offsets ? offsets[utt] : (int64_t)utt * clip_stride;
  const int len = lengths ? lengths[utt] : clip_len;
  int nf = len > 0 ? (int)((2LL * len - 1) / (2LL * fsamp)) : 0;
  if (nf > max_vf) nf = max_vf;
  const int f_lo = blockIdx.x * VAD_CHUNK;
  if (f_lo >= nf) return;
  const int16_t* x = pcm + off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec_f = fsamp >> 3;
  uint8_t* fl = flag_g + (int64_t)utt * max_vf;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    i16x8 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f_lo + wave + 4 * u;
      const bool on = f < nf && lane < nvec_f;
      v[u] = *reinterpret_cast<const i16x8*>(x + (on ? (int64_t)f * fsamp + 8 * lane : 0));
      if (!on) v[u] = (i16x8)(short)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f_lo + wave + 4 * u;
      if (f >= nf) break;  // wave-uniform
      long long acc = 0;
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        acc += (long long)((unsigned)((int)v[u][e] * (int)v[u][e]) + (unsigned)((int)v[u][e + 1] * (int)v[u][e + 1]));
      acc = wave_sum(acc);
      if (lane == 0) fl[f] = acc > threshold * (long long)fsamp ? 1 : 0;
    }
  } else {   // a clip that does not start on a 16-byte boundary: element loads
    for (int f = f_lo + wave; f < min(nf, f_lo + VAD_CHUNK); f += 4) {
      const int16_t* fr = x + (int64_t)f * fsamp;
      long long acc = 0;
      for (int i = lane; i < fsamp; i += 64) acc += (long long)((int)fr[i] * (int)fr[i]);
      acc = wave_sum(acc);
      if (lane == 0) fl[f] = acc > threshold * (long long)fsamp ? 1 : 0;
    }
  }
}

// The hysteresis of vad.py:60-129 for one clip by one wave, WITHOUT walking the frames one by one.  Between two events the
// collector's state is fully described by (triggered, s = the frame at which its ring was last cleared): at frame f the
// ring holds the frames [max(s, f - ring_len + 1), f], so with P = prefix sums of the flags every lane can evaluate "does
// the collector fire at MY frame" (voiced count > thresh when idle, unvoiced count > thresh when triggered) for 64 frames
// at once; the first lane that fires is the next event, everything before it is bulk-marked, and the scan resumes behind
// it with the new state.  Cost ~ (frames / 64 + events) wave steps instead of frames x ~40 scalar instructions (a 145 s
// clip: 4 833 frames, ~0.9 ms walked one by one).  Same decisions, same order: bit-exact with vad_kernel's phase 2.
// One wave; `flags(f)` reads frame f's flag; P: int32[nf + 1] and segm: int16[nf] scratch in LDS; returns the number of kept
// frames; on return segm[f] = segment of a kept frame or -1, and P is free again.
template <class FlagFn>
__device__ __forceinline__ int vad_walk_wave(int nf, int ring_len, int ring_thresh, int lane, FlagFn flags, int32_t* P, int16_t* segm) {
  int run = 0;
  if (lane == 0) P[0] = 0;
  for (int f0 = 0; f0 < nf; f0 += 64) {
    const int f = f0 + lane;
    const unsigned long long m = __ballot(f < nf && flags(f));
    if (f < nf) {
      P[f + 1] = run + __popcll(m & ((2ull << lane) - 1ull));
      segm[f] = -1;
    }
    run += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // one wave: LDS operations of a wave execute in order
  __builtin_amdgcn_wave_barrier();
  bool triggered = false;
  int s = 0, seg = 0;
  int f0 = 0;                                  // scan position: frames < f0 are decided
  while (f0 < nf) {
    const int f = f0 + lane;
    bool fire = false;
    if (f < nf) {
      const int lo = max(s, f - ring_len + 1);
      const int voiced = P[f + 1] - P[lo];
      fire = triggered ? (f - lo + 1 - voiced) > ring_thresh : voiced > ring_thresh;
    }
    const unsigned long long fm = __ballot(fire);
    const int last = min(nf, f0 + 64) - 1;     // last frame of this window
    if (fm == 0ull) {
      if (triggered)
        for (int g = f0 + lane; g <= last; g += 64) segm[g] = (int16_t)seg;
      f0 = last + 1;
      continue;
    }
    const int e = f0 + __ffsll((long long)fm) - 1;   // the event frame (wave-uniform)
    if (!triggered) {
      // trigger: the whole ring is emitted (vad.py:105-106), then the collector keeps frames until it releases
      const int lo = max(s, e - ring_len + 1);
      for (int g = lo + lane; g <= e; g += 64) segm[g] = (int16_t)seg;
      triggered = true;
    } else {
      // release: frames up to and including the event frame were kept (vad.py:111-123)
      for (int g = f0 + lane; g <= e; g += 64) segm[g] = (int16_t)seg;
      triggered = false;
      ++seg;
    }
    s = e + 1;
    f0 = e + 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int kept = 0;
  for (int g0 = 0; g0 < nf; g0 += 64) {
    const int g = g0 + lane;
    kept += __popcll(__ballot(g < nf && segm[g] >= 0));
  }
  return kept;
}

__global__ __launch_bounds__(64) void vad_walk_kernel(const int32_t* __restrict__ lengths, int clip_len, int fsamp, int ring_len,
                                                      int ring_thresh, int max_vf, const uint8_t* __restrict__ flag_g,
                                                      int32_t* __restrict__ pos_g, uint8_t* __restrict__ keep_out,
                                                      int32_t* __restrict__ seg_out, int32_t* __restrict__ nvf_out,
                                                      int32_t* __restrict__ voiced_len, int32_t* __restrict__ src_frame) {
  __shared__ int32_t P[MAX_VAD_FRAMES + 1];   // P[f] = flags set in frames [0, f)
  __shared__ int16_t segm[MAX_VAD_FRAMES];    // segment of a kept frame, -1 otherwise
  const int utt = blockIdx.x, lane = threadIdx.x;
  const int len = lengths ? lengths[utt] : clip_len;
  int nf = len > 0 ? (int)((2LL * len - 1) / (2LL * fsamp)) : 0;
  if (nf > max_vf) nf = max_vf;
  const uint8_t* fl = flag_g + (int64_t)utt * max_vf;
  vad_walk_wave(nf, ring_len, ring_thresh, lane, [&](int f) { return fl[f] != 0; }, P, segm);
  // packed positions = exclusive prefix count of the kept frames
  int kept = 0;
  for (int g0 = 0; g0 < max_vf; g0 += 64) {
    const int g = g0 + lane;
    const bool k = g < nf && segm[g] >= 0;
    const unsigned long long km = __ballot(k);
    if (g < max_vf) {
      const int pf = k ? kept + __popcll(km & ((1ull << lane) - 1ull)) : -1;
      if (g < nf && pos_g) pos_g[(int64_t)utt * max_vf + g] = pf;
      if (src_frame && k) src_frame[(int64_t)utt * max_vf + pf] = g;
      keep_out[(int64_t)utt * max_vf + g] = k ? 1 : 0;
      if (seg_out) seg_out[(int64_t)utt * max_vf + g] = g < nf ? (int32_t)segm[g] : -1;
    }
    kept += __popcll(km);
  }
  if (lane == 0) {
    if (nvf_out) nvf_out[utt] = nf;
    if (voiced_len) voiced_len[utt] = kept * fsamp;
  }
}

// The one-kernel path for clips of at most VAD_SMALL frames (15 s of 30 ms frames: every 3 s benchmark clip): vad_kernel's
// phases 1 and 3 with the wave-parallel walk in between and 6 KB of LDS instead of 40 (more workgroups per CU).
constexpr int VAD_SMALL = 512;

__global__ __launch_bounds__(VAD_THREADS) void vad_small_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ lengths, int64_t clip_stride,
                                                                int clip_len, int fsamp, int ring_len, int ring_thresh,
                                                                long long threshold, int max_vf, uint8_t* __restrict__ keep_out,
                                                                int32_t* __restrict__ seg_out, int32_t* __restrict__ nvf_out,
                                                                int16_t* __restrict__ voiced, int32_t* __restrict__ voiced_len, int32_t* __restrict__ src_frame) {
  __shared__ uint8_t flag[VAD_SMALL];
  __shared__ int32_t P[VAD_SMALL + 1];
  __shared__ int16_t segm[VAD_SMALL];
  __shared__ int16_t pos[VAD_SMALL];
  const int utt = blockIdx.x;
  const int64_t off = offsets ? offsets[utt] : (int64_t)utt * clip_stride;
  const int len = lengths ? lengths[utt] : clip_len;
  int nf = len > 0 ? (int)((2LL * len - 1) / (2LL * fsamp)) : 0;
  if (nf > max_vf) nf = max_vf;                       // (max_vf <= VAD_SMALL: the host checks)
  const int16_t* x = pcm + off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = VAD_THREADS / 64;
  const int nvec_f = fsamp >> 3;                      // usual geometry only (the host checks): a frame = one vector per lane
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  constexpr int VU = 8;
  for (int f0 = wave; aligned && f0 < nf; f0 += NW * VU) {
    i16x8 v[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int f = f0 + NW * u;
      const bool on = f < nf && lane < nvec_f;
      v[u] = *reinterpret_cast<const i16x8*>(x + (on ? (int64_t)f * fsamp + 8 * lane : 0));
      if (!on) v[u] = (i16x8)(short)0;
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int f = f0 + NW * u;
      if (f >= nf) break;  // wave-uniform
      long long acc = 0;
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        acc += (long long)((unsigned)((int)v[u][e] * (int)v[u][e]) + (unsigned)((int)v[u][e + 1] * (int)v[u][e + 1]));
      acc = wave_sum(acc);
      if (lane == 0) flag[f] = acc > threshold * (long long)fsamp ? 1 : 0;
    }
  }
  for (int f = wave; !aligned && f < nf; f += NW) {
    const int16_t* fr = x + (int64_t)f * fsamp;
    long long acc = 0;
    for (int i = lane; i < fsamp; i += 64) acc += (long long)((int)fr[i] * (int)fr[i]);
    acc = wave_sum(acc);
    if (lane == 0) flag[f] = acc > threshold * (long long)fsamp ? 1 : 0;
  }
  __syncthreads();
  if (wave == 0) {
    vad_walk_wave(nf, ring_len, ring_thresh, lane, [&](int f) { return flag[f] != 0; }, P, segm);
    int kept = 0;
    for (int g0 = 0; g0 < max_vf; g0 += 64) {
      const int g = g0 + lane;
      const bool k = g < nf && segm[g] >= 0;
      const unsigned long long km = __ballot(k);
      if (g < max_vf) {
        const int pf_s = k ? kept + __popcll(km & ((1ull << lane) - 1ull)) : -1;
        if (g < nf) pos[g] = (int16_t)pf_s;
        if (src_frame && k) src_frame[(int64_t)utt * max_vf + pf_s] = g;
        keep_out[(int64_t)utt * max_vf + g] = k ? 1 : 0;
        if (seg_out) seg_out[(int64_t)utt * max_vf + g] = g < nf ? (int32_t)segm[g] : -1;
      }
      kept += __popcll(km);
    }
    if (lane == 0) {
      if (nvf_out) nvf_out[utt] = nf;
      if (voiced_len) voiced_len[utt] = kept * fsamp;
    }
  }
  __syncthreads();
  if (voiced) {
    int16_t* dst = voiced + off;
    const bool vec_copy = aligned && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int f0 = wave; vec_copy && f0 < nf; f0 += NW * VU) {  // eight frames' loads first, then their stores
      i16x8 v[VU];
      int pf[VU];
#pragma unroll
      for (int u = 0; u < VU; ++u) {
        const int f = f0 + NW * u;
        pf[u] = f < nf ? (int)pos[f] : -1;
        if (pf[u] >= 0 && lane < nvec_f) v[u] = *reinterpret_cast<const i16x8*>(x + (int64_t)f * fsamp + 8 * lane);
      }
#pragma unroll
      for (int u = 0; u < VU; ++u)
        if (pf[u] >= 0 && lane < nvec_f) *reinterpret_cast<i16x8*>(dst + (int64_t)pf[u] * fsamp + 8 * lane) = v[u];
    }
    for (int f = wave; !vec_copy && f < nf; f += NW) {
      const int pf = pos[f];
      if (pf < 0) continue;
      for (int i = lane; i < fsamp; i += 64) dst[(int64_t)pf * fsamp + i] = x[(int64_t)f * fsamp + i];
    }
  }
}

__global__ __launch_bounds__(VAD_THREADS) void vad_copy_kernel(const int16_t* __restrict__ pcm, const int64_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ lengths, int64_t clip_stride, int clip_len,
                                                               int fsamp, int max_vf, const int32_t* __restrict__ pos_g,
                                                               int16_t* __restrict__ voiced) {
  const int utt = blockIdx.y;
  const int64_t off = offsets ? offsets[utt] : (int64_t)utt * clip_stride;
  const int len = lengths ? lengths[utt] : clip_len;
  int nf = len > 0 ? (int)((2LL * len - 1) / (2LL * fsamp)) : 0;
  if (nf > max_vf) nf = max_vf;
  const int f_lo = blockIdx.x * VAD_CHUNK;
  if (f_lo >= nf) return;
  const int16_t* x = pcm + off;
  int16_t* dst = voiced + off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec_f = fsamp >> 3;
  const int32_t* pp = pos_g + (int64_t)utt * max_vf;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    i16x8 v[8];
    int pf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f_lo + wave + 4 * u;
      pf[u] = f < nf ? pp[f] : -1;
      if (pf[u] >= 0 && lane < nvec_f) v[u] = *reinterpret_cast<const i16x8*>(x + (int64_t)f * fsamp + 8 * lane);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (pf[u] >= 0 && lane < nvec_f) *reinterpret_cast<i16x8*>(dst + (int64_t)pf[u] * fsamp + 8 * lane) = v[u];
  } else {
    for (int f = f_lo + wave; f < min(nf, f_lo + VAD_CHUNK); f += 4) {
      const int pf = pp[f];
      if (pf < 0) continue;
      for (int i = lane; i < fsamp; i += 64) dst[(int64_t)pf * fsamp + i] = x[(int64_t)f * fsamp + i];
    }
  }
}

}  // namespace

extern "C" int svk_vad_energy(svk_ctx* ctx, const int16_t* d_pcm, const int64_t* d_offsets, const int32_t* d_lengths,
                              int64_t clip_stride, int32_t clip_len, int32_t n_utt, int32_t frame_samples,
                              int32_t ring_len, int32_t ring_thresh, int64_t threshold, int32_t max_vad_frames,
                              uint8_t* d_keep, int32_t* d_seg, int32_t* d_n_vad_frames, int16_t* d_voiced,
                              int32_t* d_voiced_len, int32_t* d_src_frame) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && frame_samples >= 1 && ring_len >= 1 && ring_thresh >= 0 && max_vad_frames >= 0,
              "negative or zero geometry");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_pcm && d_keep, "d_pcm / d_keep is NULL");
  SVK_REQUIRE(ctx, !d_voiced || d_voiced_len, "d_voiced needs d_voiced_len");
  SVK_REQUIRE(ctx, !d_src_frame || d_voiced_len, "d_src_frame needs d_voiced_len");
  SVK_REQUIRE(ctx, d_voiced != d_pcm, "d_voiced must not alias d_pcm");
  if (max_vad_frames > MAX_VAD_FRAMES)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "at most %d VAD frames per clip, got %d", MAX_VAD_FRAMES,
                    max_vad_frames);
  // long clips: the three phases as three kernels cut into frame chunks (see vad_flags_kernel); SVK_VAD_SPLIT=0 / 1 forces a path
  const char* force = getenv("SVK_VAD_SPLIT");
  const bool usual = (frame_samples & 7) == 0 && (frame_samples >> 3) <= 64;
  const bool split = usual && n_utt <= 65535 && (force ? force[0] == '1' : max_vad_frames > 512);
  if (split && max_vad_frames > 0) {
    const size_t flag_bytes = ((size_t)n_utt * max_vad_frames + 15) & ~(size_t)15;
    const size_t pos_bytes = d_voiced ? sizeof(int32_t) * (size_t)n_utt * max_vad_frames : 0;   // positions: only the copy reads them
    const int rc = svk_ensure_work(ctx, flag_bytes + pos_bytes);
    if (rc != SVK_OK) return rc;
    uint8_t* flag_g = reinterpret_cast<uint8_t*>(ctx->work);
    int32_t* pos_g = d_voiced ? reinterpret_cast<int32_t*>(flag_g + flag_bytes) : nullptr;
    const dim3 grid((max_vad_frames + VAD_CHUNK - 1) / VAD_CHUNK, n_utt);
    hipLaunchKernelGGL(vad_flags_kernel, grid, dim3(VAD_THREADS), 0, ctx->stream, d_pcm, d_offsets, d_lengths, clip_stride, clip_len,
                       frame_samples, (long long)threshold, max_vad_frames, flag_g);
    SVK_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(vad_walk_kernel, dim3(n_utt), dim3(64), 0, ctx->stream, d_lengths, clip_len, frame_samples, ring_len,
                       ring_thresh, max_vad_frames, flag_g, pos_g, d_keep, d_seg, d_n_vad_frames, d_voiced_len, d_src_frame);
    SVK_LAUNCH_CHECK(ctx);
    if (d_voiced) {
      hipLaunchKernelGGL(vad_copy_kernel, grid, dim3(VAD_THREADS), 0, ctx->stream, d_pcm, d_offsets, d_lengths, clip_stride, clip_len,
                         frame_samples, max_vad_frames, pos_g, d_voiced);
      SVK_LAUNCH_CHECK(ctx);
    }
    return SVK_OK;
  }
  if (usual && max_vad_frames <= VAD_SMALL && !(force && force[0] == '2')) {   // (SVK_VAD_SPLIT=2 forces the general one-kernel path)
    hipLaunchKernelGGL(vad_small_kernel, dim3(n_utt), dim3(VAD_THREADS), 0, ctx->stream, d_pcm, d_offsets, d_lengths, clip_stride,
                       clip_len, frame_samples, ring_len, ring_thresh, (long long)threshold, max_vad_frames, d_keep, d_seg,
                       d_n_vad_frames, d_voiced, d_voiced_len, d_src_frame);
    SVK_LAUNCH_CHECK(ctx);
    return SVK_OK;
  }
  hipLaunchKernelGGL(vad_kernel, dim3(n_utt), dim3(VAD_THREADS), 0, ctx->stream, d_pcm, d_offsets, d_lengths, clip_stride,
                     clip_len, frame_samples, ring_len, ring_thresh, (long long)threshold, max_vad_frames, d_keep,
                     d_seg, d_n_vad_frames, d_voiced, d_voiced_len, d_src_frame);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}
