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
                                                  int16_t* __restrict__ voiced, int32_t* __restrict__ voiced_len) {
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

}  // namespace

extern "C" int svk_vad_energy(svk_ctx* ctx, const int16_t* d_pcm, const int64_t* d_offsets, const int32_t* d_lengths,
                              int64_t clip_stride, int32_t clip_len, int32_t n_utt, int32_t frame_samples,
                              int32_t ring_len, int32_t ring_thresh, int64_t threshold, int32_t max_vad_frames,
                              uint8_t* d_keep, int32_t* d_seg, int32_t* d_n_vad_frames, int16_t* d_voiced,
                              int32_t* d_voiced_len) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0 && frame_samples >= 1 && ring_len >= 1 && ring_thresh >= 0 && max_vad_frames >= 0,
              "negative or zero geometry");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_pcm && d_keep, "d_pcm / d_keep is NULL");
  SVK_REQUIRE(ctx, !d_voiced || d_voiced_len, "d_voiced needs d_voiced_len");
  SVK_REQUIRE(ctx, d_voiced != d_pcm, "d_voiced must not alias d_pcm");
  if (max_vad_frames > MAX_VAD_FRAMES)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "at most %d VAD frames per clip, got %d", MAX_VAD_FRAMES,
                    max_vad_frames);
  hipLaunchKernelGGL(vad_kernel, dim3(n_utt), dim3(VAD_THREADS), 0, ctx->stream, d_pcm, d_offsets, d_lengths, clip_stride,
                     clip_len, frame_samples, ring_len, ring_thresh, (long long)threshold, max_vad_frames, d_keep,
                     d_seg, d_n_vad_frames, d_voiced, d_voiced_len);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}
