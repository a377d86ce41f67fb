// The end of the C3D2 embedding network on v_mfma_f32_16x16x4_f32 (model.py:136-139 definitions, :167-170 forward; conv4_1
// runs on the f16 matrix pipe in csrc/c3d2.hip, where conv3_2 went before it -- both were instances of this template first:
// tools/experiments/conv32_f32_template_instance.patch, conv41_f32_template_instance.patch):
//   c3d2_tail_kernel<Conv42>   conv4_2 (128 -> 128, kernel (3,7,1)) + BN + PReLU
//   fc5_kernel / fc5_reduce_kernel   FC5 (4 608 -> 128), K split four ways, partial sums added in a fixed order
// BatchNorm (eval mode) is folded into weights and biases by the host (model.FusedEmbedder).
//
// These layers are GEMM-shaped over the BATCH: per cube they have 36 / 1 output positions but K = 2 688 / 4 608 and
// N = 128, and their weights (1.7 / 2.4 MB) fit no register file.  conv4_2 is 3 taps deep with depth stride 1, so
// Winograd's F(2, 3) along depth applies: for an output depth pair (2 P, 2 P + 1),
//     t0 = x0 - x2,  t1 = x1 + x2,  t2 = x2 - x1,  t3 = x1 - x3            (input depths x0 .. x3 = 2 P .. 2 P + 3)
//     a_k = sum over (row / column tap, input channel) of t_k G_k          (G: transformed weights, made by the HOST here)
//     y(2 P) = a0 + a1 + a2,   y(2 P + 1) = a1 - a2 - a3                   (4 MFMAs where the direct form issues 6)
//
// Shape of the convolution kernel (a template over the layer's geometry; conv4_2 is the instance left):
//   * M tile = ONE output position of SIXTEEN cubes (lane i = cube): every tile is full whatever the layer's 9 or 27
//     positions per depth pair -- tiles cut inside a cube would be 27 of 32 and 9 of 16 rows full;
//   * work item = (group of 16 cubes, depth pair[, block of rows]) = 9 positions = 9 M tiles; the workgroup's
//     eight waves own one 16-channel N tile each: 9 tiles x 4 transformed accumulators = 144 VGPRs, two waves per SIMD;
//   * K runs over (8-channel chunk, tap, k).  A chunk of the item's input is staged in LDS ALREADY TRANSFORMED: the
//     staging threads load the four depths of a (cube, pixel, 4 channels) piece, form t0 .. t3 once and park them as
//     t planes [cube][k][pixel][8]; the main loop is then ds_read_b64 fragments + MFMAs and nothing else (the kernels of
//     c3d2.hip transform per fragment read: 8 packed adds per 16 MFMAs -- VALU that f32 MFMAs never overlap with);
//   * two LDS buffers: the next chunk is fetched, transformed and parked in two rounds of 16-byte pieces inside the
//     current chunk's matrix work; one barrier per chunk;
//   * the B operand never touches LDS: a wave streams ITS N tile's fragments [chunk][tap][k][64 lanes][2] linearly from
//     global memory (L2-resident: every workgroup reads the same 0.4 / 1.8 MB), one 512-byte load per 18 MFMAs, three deep;
//   * activations between these layers use a CHUNKED layout [cube][depth][chunk of 8 channels][pixel][8] so that a staged
//     chunk is contiguous in memory (channels-last would serve 32-byte pieces of 256 / 512-byte pixels).
// LDS: a cube's chunk sits at a stride = 4 (mod 64) floats: the 32 lanes of a ds_read_b64 group (16 cubes x 2 K pairs)
// then fall into 32 different 8-byte slots -- conflict-free by the lane-group table of MI355X_MICROARCH.md (LDS).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "svk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float prelu(float v, float slope) { return v > 0.f ? v : slope * v; }
template <bool SLOPE01>
__device__ __forceinline__ float prelu_t(float v, float slope) {
  return SLOPE01 ? fmaxf(v, slope * v) : prelu(v, slope);
}

constexpr int GROUP = 16;   // cubes per work item = rows of an M tile

// conv4_2: input = conv4_1's output, output [n][4 d][16 chunks][9 = 3 h x 3 w][8].  Item = (group, pair P of 2), taps along h.
struct Conv42 {
  static constexpr int NT = 9;
  static constexpr int TAPS = 7, TAP_PIX = 3;          // a tap moves one row = 3 pixels
  static constexpr int D_IN = 6, NCHUNK = 16, PIX_IN = 27, PIXN = 27;
  static constexpr int D_OUT = 4, PIX_OUT = 9;
  static constexpr int SC = 1;
  static constexpr int PAIRS = 2, BLOCKS = 1;
  static constexpr int NWAVES = 8;
  __device__ static constexpr int pix0(int p) { return p; }
  __device__ static int in_pix_start(int) { return 0; }
  __device__ static int out_pix(int t, int) { return t; }
};
// (conv3_2 ran as a third instance of this template in round 3 and the first half of round 4; it is c3d2_conv32h_kernel in c3d2.hip now.)
template <class L>
struct TailGeom {
  static constexpr int PLANE = L::PIXN * 8;                                   // floats per t plane of one cube and chunk
  static constexpr int CS = 4 * PLANE + ((4 - (4 * PLANE) % 64) + 64) % 64;   // cube stride = 4 (mod 64) floats
  static constexpr int SUB = GROUP * CS;                                      // floats per staged chunk
  static constexpr int BUF = L::SC * SUB;                                     // floats per LDS buffer
  static constexpr int THREADS = 64 * L::NWAVES;
  static constexpr int OUT_CHUNKS = 2 * L::NWAVES;                            // 8-channel chunks of the output
  static constexpr int UNITS = L::SC * GROUP * L::PIXN * 2;                   // 16-byte (cube, pixel, half chunk) pieces per phase
  static constexpr int ROUNDS = (UNITS + THREADS - 1) / THREADS;
  static constexpr int NPH = L::NCHUNK / L::SC;                               // phases per item
  static constexpr int STEPS = L::SC * L::TAPS * 4;                           // (chunk, tap, k) steps per phase
  static constexpr int ITEMS_PER_GROUP = L::PAIRS * L::BLOCKS;
  static constexpr int64_t IN_CUBE = (int64_t)L::D_IN * L::NCHUNK * L::PIX_IN * 8;
  static constexpr int64_t OUT_CUBE = (int64_t)L::D_OUT * OUT_CHUNKS * L::PIX_OUT * 8;
  static_assert(CS % 64 == 4, "cube stride");
  static_assert(ROUNDS == 2, "the staging schedule below is written for two rounds");
  static_assert(SUB * 4 <= 65532, "fragment offsets inside a staged chunk must fit the DS instruction's 16-bit field");
};

struct TailParams {
  const float* in;
  const f32x2* wfrag;   // [nt][NCHUNK][TAPS][4 k][64 lanes]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: G_k[co][8 chunk + 2 kk + e][tap]
  const float* bias;    // [16 x N tiles]
  const float* slope;
  float* out;
  int32_t n_utt;
  unsigned* queue;      // work-item counter (zeroed before the launch) where workgroups share a CU, or NULL = fixed stride
  unsigned long long* stamps;   // tuning builds only (-DSVK_TUNING): [grid][waves][4] summed cycles
};

#ifdef SVK_TUNING
#define TAIL_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define TAIL_STAMP_ADD(slot, a, b) do { stamp_acc[slot] += (b) - (a); } while (0)
#else
#define TAIL_STAMP(var) do { } while (0)
#define TAIL_STAMP_ADD(slot, a, b) do { } while (0)
#endif

template <class L, bool SLOPE01>
__global__ __launch_bounds__(64 * L::NWAVES, 2) void c3d2_tail_kernel(const TailParams p) {
  using G = TailGeom<L>;
  constexpr int TAIL_THREADS = G::THREADS;
  constexpr int NT = L::NT;
  extern __shared__ __attribute__((aligned(16))) float smem_tail[];
  __shared__ int q_next;
  const int lane = threadIdx.x & 63, nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  // The MFMA's M side is the CHANNEL (A = the streamed weight fragment), its N side the cube (B = the staged activations): a lane
  // ends up with channels 16 nt + 4 kk .. + 3 of cube i -- 16 contiguous bytes of the chunked output, one 16-byte store per
  // (tile, depth) where the other order (M = cube) issued four 4-byte stores (round 4: a quarter of the store instructions; the same products
  // in the same order: bit-identical)
  f32x4 bias4, slope4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bias4[r] = p.bias[16 * nt + 4 * kk + r];
    slope4[r] = p.slope[16 * nt + 4 * kk + r];
  }
  const int n_groups = (p.n_utt + GROUP - 1) / GROUP;
  const int n_items = n_groups * G::ITEMS_PER_GROUP;

  // ---- staging: thread -> (round, chunk of the phase, cube, pixel, half) pieces; constant over the kernel ----
  int st_goff[G::ROUNDS], st_loff[G::ROUNDS], st_cube[G::ROUNDS];
#pragma unroll
  for (int r = 0; r < G::ROUNDS; ++r) {
    const int u = threadIdx.x + TAIL_THREADS * r;
    const int uu = u < G::UNITS ? u : 0;
    const int sub = uu / (GROUP * L::PIXN * 2), rem = uu - sub * (GROUP * L::PIXN * 2);
    const int cube = rem / (L::PIXN * 2), r2 = rem - cube * (L::PIXN * 2);
    st_cube[r] = u < G::UNITS ? cube : -1;
    st_goff[r] = sub * (L::PIX_IN * 8) + 4 * r2;                    // + cube term (clamped per item) + item / phase base
    st_loff[r] = sub * G::SUB + cube * G::CS + 4 * r2;              // pixel r2 >> 1, half r2 & 1: 8 (r2 >> 1) + 4 (r2 & 1) = 4 r2
  }
  const f32x2* const wb = p.wfrag + (size_t)nt * (L::NCHUNK * L::TAPS * 4 * 64) + lane;
  const int a_lane = i * G::CS + 2 * kk;                            // this lane's part of every fragment address

  f32x4 sx[4];                                                       // one round of staged pieces: the four depths
  auto item_base = [&](int item, int ph, const float*& src, int& cube_lim) {
    const int g = item / G::ITEMS_PER_GROUP, rem = item - g * G::ITEMS_PER_GROUP;
    const int P = rem / L::BLOCKS, blk = rem - P * L::BLOCKS;
    cube_lim = min(GROUP, p.n_utt - GROUP * g) - 1;                 // cubes past the batch re-read the group's last one
    src = p.in + (int64_t)g * GROUP * G::IN_CUBE + ((int64_t)(2 * P) * L::NCHUNK + ph * L::SC) * (L::PIX_IN * 8) +
          L::in_pix_start(blk) * 8;
  };
  auto stage_load = [&](const float* src, int cube_lim, int r) {
    if (st_cube[r] >= 0) {
      const float* s = src + (int64_t)min(st_cube[r], cube_lim) * G::IN_CUBE + st_goff[r];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) sx[dd] = *reinterpret_cast<const f32x4*>(s + dd * (L::NCHUNK * L::PIX_IN * 8));
    }
  };
  auto stage_park = [&](float* buf, int r) {
    if (st_cube[r] >= 0) {
      float* d = buf + st_loff[r];
      *reinterpret_cast<f32x4*>(d) = sx[0] - sx[2];
      *reinterpret_cast<f32x4*>(d + G::PLANE) = sx[1] + sx[2];
      *reinterpret_cast<f32x4*>(d + 2 * G::PLANE) = sx[2] - sx[1];
      *reinterpret_cast<f32x4*>(d + 3 * G::PLANE) = sx[1] - sx[3];
    }
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  {
    const float* src;
    int lim;
    item_base(item, 0, src, lim);
#pragma unroll
    for (int r = 0; r < G::ROUNDS; ++r) {
      stage_load(src, lim, r);
      stage_park(smem_tail, r);
    }
  }
  __syncthreads();

  f32x4 acc[NT][4];
  int buf_sel = 0;
  f32x2 b0 = wb[0], b1 = wb[64];
  // the item after this one: a fixed stride, or (workgroups sharing a CU: see c3d2_conv21w_kernel) a ticket of the device-wide
  // counter drawn one item ahead -- the staging of an item's last phase already needs to know its successor
  int item_next = item + (int)gridDim.x;
  if (p.queue) {
    if (threadIdx.x == 0) q_next = (int)atomicAdd(p.queue, 1u) + (int)gridDim.x;
    __syncthreads();
    item_next = q_next;
  }
#ifdef SVK_TUNING
  unsigned long long stamp_acc[4] = {0, 0, 0, 0};
#endif
  while (item < n_items) {
    unsigned q_ticket = 0;
    if (p.queue && threadIdx.x == 0) q_ticket = atomicAdd(p.queue, 1u);   // for the item after next; published below
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[t][k] = k == 1 ? bias4 : (f32x4){0.f, 0.f, 0.f, 0.f};   // a1 carries the bias
#pragma unroll 1
    for (int ph = 0; ph < G::NPH; ++ph) {
      TAIL_STAMP(tp0);
      // what the NEXT phase stages (the next item's first phase behind this item's last)
      const bool last_ph = ph + 1 == G::NPH;
      const int n_item = last_ph ? item_next : item, n_ph = last_ph ? 0 : ph + 1;
      const bool have_next = n_item < n_items;
      const float* nsrc = p.in;
      int nlim = 0;
      if (have_next) item_base(n_item, n_ph, nsrc, nlim);
      const float* const abuf = smem_tail + buf_sel * G::BUF + a_lane;
      float* const nbuf = smem_tail + (buf_sel ^ 1) * G::BUF;
      const f32x2* const wph = wb + (size_t)ph * (G::STEPS * 64);
      const f32x2* const wnph = wb + (size_t)n_ph * (G::STEPS * 64);   // the stream wraps to the item's start behind its last phase

      // B fragments run two steps ahead of their use (b0: this step, b1: the next; across phases and items), A
      // fragments one step ahead
      f32x2 a[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const f32x2*>(abuf + L::pix0(t) * 8);
#pragma unroll
      for (int s = 0; s < G::STEPS; ++s) {
        const int k = s & 3;
        const f32x2 b2 = s + 2 < G::STEPS ? wph[(s + 2) * 64] : wnph[(s + 2 - G::STEPS) * 64];
        // the next phase's chunk: fetched, transformed and parked in two rounds inside this phase's matrix work
        if (have_next) {
          if (s == 1) stage_load(nsrc, nlim, 0);
          if (s == G::STEPS / 2 - 2) stage_park(nbuf, 0);
          if (s == G::STEPS / 2 - 1) stage_load(nsrc, nlim, 1);
          if (s == G::STEPS - 3) stage_park(nbuf, 1);
        }
        __builtin_amdgcn_sched_barrier(0);   // those loads are issued in front of this step's MFMAs
        const f32x2 bv = b0;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0], a[t][0], acc[t][k], 0, 0, 0);
        // second K pair; tile t's fragment of the NEXT step is read as soon as this step's last MFMA on it has issued
        // (nine MFMAs = 288 cycles before its first use: no second fragment set in registers)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1], a[t][1], acc[t][k], 0, 0, 0);
          if (s + 1 < G::STEPS) {
            const int s1 = s + 1, k1 = s1 & 3, tap1 = (s1 >> 2) % L::TAPS, sub1 = (s1 >> 2) / L::TAPS;
            __builtin_amdgcn_sched_barrier(0);
            a[t] = *reinterpret_cast<const f32x2*>(abuf + sub1 * G::SUB + k1 * G::PLANE + (L::pix0(t) + tap1 * L::TAP_PIX) * 8);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        b0 = b1;
        b1 = b2;
      }
      TAIL_STAMP(tp1);
      __syncthreads();   // the next chunk is parked; this one may be overwritten by the phase after next
      TAIL_STAMP(tp2);
      TAIL_STAMP_ADD(0, tp0, tp1);   // a phase: steps of 18 MFMAs + the next chunk's staging
      TAIL_STAMP_ADD(1, tp1, tp2);   // its barrier
      buf_sel ^= 1;
    }
    TAIL_STAMP(te0);
    int item_after = item_next + (int)gridDim.x;
    if (p.queue) {
      // (the phase loop above ended with a barrier: q_next's previous value has been read by every thread)
      if (threadIdx.x == 0) q_next = (int)q_ticket + (int)gridDim.x;
    }
    // ---- output transform, PReLU, stores: rows 4 kk + r = channel 16 nt + 4 kk + r, column i = cube ----
    {
      const int g = item / G::ITEMS_PER_GROUP, rem = item - g * G::ITEMS_PER_GROUP;
      const int P = rem / L::BLOCKS, blk = rem - P * L::BLOCKS;
      const int n_here = min(GROUP, p.n_utt - GROUP * g);
      // chunked output: [cube][depth][chunk = 2 nt + (kk >> 1)][pixel][4 (kk & 1) .. + 3]
      float* const o = p.out + (int64_t)g * GROUP * G::OUT_CUBE +
                       ((int64_t)(2 * P) * G::OUT_CHUNKS + 2 * nt) * (L::PIX_OUT * 8);   // wave-uniform
      const int olane = i * (int)G::OUT_CUBE + (kk >> 1) * (L::PIX_OUT * 8) + 4 * (kk & 1);
      if (i < n_here) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 y0 = acc[t][0] + acc[t][1] + acc[t][2], y1 = acc[t][1] - acc[t][2] - acc[t][3];
          const int opix = L::out_pix(t, blk) * 8;
          f32x4 o0, o1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o0[r] = prelu_t<SLOPE01>(y0[r], slope4[r]);
            o1[r] = prelu_t<SLOPE01>(y1[r], slope4[r]);
          }
          *reinterpret_cast<f32x4*>(o + olane + opix) = o0;
          *reinterpret_cast<f32x4*>(o + olane + opix + G::OUT_CHUNKS * L::PIX_OUT * 8) = o1;
        }
      }
    }
    item = item_next;
    if (p.queue) {
      __syncthreads();               // q_next (written before the epilogue) is visible
      item_next = q_next;
    } else {
      item_next = item_after;
    }
    TAIL_STAMP(te1);
    TAIL_STAMP_ADD(2, te0, te1);     // output transform, PReLU, stores (+ the queue barrier)
#ifdef SVK_TUNING
    stamp_acc[3] += 1;               // items this workgroup processed
#endif
  }
#ifdef SVK_TUNING
  if (p.stamps && lane == 0)
    for (int k = 0; k < 4; ++k) p.stamps[((size_t)blockIdx.x * L::NWAVES + nt) * 4 + k] = stamp_acc[k];
#endif
}

template <class L>
int launch_tail(svk_ctx* ctx, const char* name, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                const float* d_slope, int32_t flags, float* d_out) {
  using G = TailGeom<L>;
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  SVK_REQUIRE(ctx, (flags & ~2) == 0, "flags: only bit 1 (slopes in [0, 1]) is defined");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wfrag && d_bias && d_slope && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wfrag)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt < ((int64_t)1 << 24), "too many cubes for one launch");
  const size_t lds = sizeof(float) * (size_t)(2 * G::BUF);
  if (lds > (size_t)ctx->lds_per_cu)
    return svk_fail(ctx, SVK_ERR_UNSUPPORTED, "%s needs %zu bytes of LDS per workgroup (device: %d)", name, lds, ctx->lds_per_cu);
  void (*kern)(const TailParams) = (flags & 2) ? c3d2_tail_kernel<L, true> : c3d2_tail_kernel<L, false>;
  SVK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t items = (int64_t)((n_utt + GROUP - 1) / GROUP) * G::ITEMS_PER_GROUP;
  int per_cu = 1;
  if (L::NWAVES < 8) {   // four-wave workgroups share a CU
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), G::THREADS, lds) != hipSuccess || per_cu < 1)
      per_cu = 1;
    per_cu = std::min(per_cu, 2);
  }
  unsigned* queue = nullptr;
  if (per_cu > 1 && !getenv("SVK_C3D2_STATIC_ITEMS")) {   // a slot of the handle's 256-byte scratch, zeroed in stream order
    queue = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->scratch) + 96);
    SVK_HIP(ctx, hipMemsetAsync(queue, 0, 4, ctx->stream));
  }
  TailParams p{d_in, reinterpret_cast<const f32x2*>(d_wfrag), d_bias, d_slope, d_out, n_utt, queue, nullptr};
  const unsigned grid = (unsigned)std::min<int64_t>(items, (int64_t)per_cu * ctx->num_cu);
#ifdef SVK_TUNING
  const bool want_stamps = getenv("SVK_C3D2_STAMPS") != nullptr;
  const size_t stamp_bytes = (size_t)grid * L::NWAVES * 4 * sizeof(unsigned long long);
  if (want_stamps) {
    const int rc = svk_ensure_work(ctx, stamp_bytes);
    if (rc != SVK_OK) return rc;
    p.stamps = reinterpret_cast<unsigned long long*>(ctx->work);
    SVK_HIP(ctx, hipMemsetAsync(p.stamps, 0, stamp_bytes, ctx->stream));
  }
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::THREADS), lds, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
#ifdef SVK_TUNING
  if (want_stamps) {   // cycles per ITEM (s_memtime), averaged over workgroups, wave 0 and the last wave
    std::vector<unsigned long long> h((size_t)grid * L::NWAVES * 4);
    SVK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SVK_HIP(ctx, hipMemcpy(h.data(), p.stamps, stamp_bytes, hipMemcpyDeviceToHost));
    for (int w : {0, L::NWAVES - 1}) {
      double s0 = 0, s1 = 0, s2 = 0, n = 0;
      for (unsigned b = 0; b < grid; ++b) {
        const unsigned long long* e = &h[((size_t)b * L::NWAVES + w) * 4];
        s0 += (double)e[0]; s1 += (double)e[1]; s2 += (double)e[2]; n += (double)e[3];
      }
      fprintf(stderr, "%s stamps wave %d (cycles per item; %d phases of %d steps x %d MFMAs = %d MFMA cycles per wave): phases %.0f  barriers %.0f  epilogue %.0f\n",
              name, w, G::NPH, G::STEPS, 2 * L::NT, G::NPH * G::STEPS * 2 * L::NT * 32, s0 / n, s1 / n, s2 / n);
    }
  }
#endif
  return SVK_OK;
}

// ---- FC5 (4 608 -> 128, model.py:136 + :168-169; no BatchNorm, no activation behind it) on the output of conv4_2 in its
// chunked layout: row u = 4 608 floats ordered (d, chunk, pixel, channel % 8) -- the host permutes FC5's columns to match.
// M = the batch, so an M tile is 16 cubes; a workgroup takes 64 cubes (four tiles: a B fragment feeds 16 MFMAs -- with one
// tile every workgroup would stream all 2.4 MB of weights for 16 cubes) and ONE of the four depths d = a K range of
// 1 152: partial sums [4 d][n][128], added in the fixed order d = 0 .. 3 by fc5_reduce_kernel (no atomics: the embeddings
// stay bitwise repeatable).  Wave = N tile; A rows staged through two LDS buffers of 64 x 128 floats (row stride 136 = 8
// mod 64: conflict-free ds_read_b128 by the lane-group table), B fragments [d][nt][72][64 lanes][4] straight from global. ----
constexpr int TAIL_THREADS = 512;
constexpr int FC_K = 4608, FC_KD = 1152, FC_CUBES = 64, FC_CH = 128, FC_ROW = 136;

struct Fc5Params {
  const float* in;      // [n][4608]
  const f32x4* wfrag;   // [4 d][8 nt][72][64]: lane (j = 16 nt + (l & 15), kk = l >> 4), e: W5[j][column of K index 1152 d + 16 step + 4 kk + e]
  float* part;          // [4 d][n][128]
  int32_t n_utt;
};

__global__ __launch_bounds__(TAIL_THREADS) void fc5_kernel(const Fc5Params p) {
  __shared__ __attribute__((aligned(16))) float rows[2][FC_CUBES * FC_ROW];
  const int lane = threadIdx.x & 63, nt = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int n_groups = (p.n_utt + FC_CUBES - 1) / FC_CUBES;
  // staging: thread -> row (cube) t / 8 + 64 j ... : 2 048 sixteen-byte pieces per chunk, four per thread
  const int s_row = threadIdx.x >> 5, s_piece = threadIdx.x & 31;           // rows s_row + 16 q, q < 4
  for (int item = blockIdx.x; item < n_groups * 4; item += gridDim.x) {
    const int g = item >> 2, d = item & 3;
    const int n_here = min(FC_CUBES, p.n_utt - FC_CUBES * g);
    const float* src = p.in + (int64_t)g * FC_CUBES * FC_K + d * FC_KD + 4 * s_piece;
    const f32x4* wb = p.wfrag + ((size_t)(d * 8 + nt) * 72) * 64 + lane;
    f32x4 pre[4];
    auto fetch = [&](int c) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pre[q] = *reinterpret_cast<const f32x4*>(src + (int64_t)min(s_row + 16 * q, n_here - 1) * FC_K + c * FC_CH);
    };
    auto park = [&](int b) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&rows[b][(s_row + 16 * q) * FC_ROW + 4 * s_piece]) = pre[q];
    };
    __syncthreads();   // the previous item's readers are done with both buffers
    fetch(0);
    park(0);
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 b0 = wb[0], b1 = wb[64];
#pragma unroll 1
    for (int c = 0; c < FC_KD / FC_CH; ++c) {          // nine chunks of 128 K values = eight 16-float steps each
      if (c + 1 < FC_KD / FC_CH) fetch(c + 1);
      const float* ab = &rows[c & 1][i * FC_ROW + 4 * kk];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int st = 8 * c + j;
        const f32x4 b2 = wb[(size_t)min(st + 2, 71) * 64];
        f32x4 a[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(ab + 16 * mt * FC_ROW + 16 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][e], b0[e], acc[mt], 0, 0, 0);
        b0 = b1;
        b1 = b2;
      }
      if (c + 1 < FC_KD / FC_CH) park((c + 1) & 1);     // its last readers finished before the barrier of chunk c - 1
      __syncthreads();
    }
    // rows 4 kk + r of tile mt = cube 64 g + 16 mt + 4 kk + r, column i = output 16 nt + i
    float* const o = p.part + ((int64_t)d * p.n_utt + (int64_t)FC_CUBES * g) * 128 + 16 * nt + i;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * kk + r;
        if (row < n_here) o[row * 128] = acc[mt][r];
      }
  }
}

__global__ __launch_bounds__(256) void fc5_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t n_vec, int64_t plane) {
  // out[u][j] = ((part0 + part1) + part2) + part3 + bias[j], four floats per thread
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 4 * (v & 31));
    const f32x4* q = reinterpret_cast<const f32x4*>(part) + v;
    const int64_t pv = plane / 4;
    reinterpret_cast<f32x4*>(out)[v] = (((q[0] + q[pv]) + q[2 * pv]) + q[3 * pv]) + b;
  }
}

}  // namespace

extern "C" size_t svk_c3d2_fc5_workspace_floats(int32_t n_utt) { return (size_t)4 * (size_t)(n_utt > 0 ? n_utt : 0) * 128; }

extern "C" int svk_c3d2_fc5(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                            float* d_work, float* d_out) {
  if (!ctx) return SVK_ERR_BAD_ARG;
  SVK_REQUIRE(ctx, n_utt >= 0, "n_utt negative");
  if (n_utt == 0) return SVK_OK;
  SVK_REQUIRE(ctx, d_in && d_wfrag && d_bias && d_work && d_out, "NULL buffer");
  SVK_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_wfrag) | reinterpret_cast<uintptr_t>(d_work) |
                     reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_bias)) & 15) == 0,
              "buffers must be 16-byte aligned");
  SVK_REQUIRE(ctx, (int64_t)n_utt < ((int64_t)1 << 24), "too many cubes for one launch");
  Fc5Params p{d_in, reinterpret_cast<const f32x4*>(d_wfrag), d_work, n_utt};
  const int64_t items = (int64_t)((n_utt + FC_CUBES - 1) / FC_CUBES) * 4;
  hipLaunchKernelGGL(fc5_kernel, dim3((unsigned)std::min<int64_t>(items, ctx->num_cu)), dim3(TAIL_THREADS), 0, ctx->stream, p);
  SVK_LAUNCH_CHECK(ctx);
  const int64_t n_vec = (int64_t)n_utt * 32;
  hipLaunchKernelGGL(fc5_reduce_kernel, dim3((unsigned)std::min<int64_t>((n_vec + 255) / 256, (int64_t)ctx->num_cu * 4)), dim3(256),
                     0, ctx->stream, d_work, d_bias, d_out, n_vec, (int64_t)n_utt * 128);
  SVK_LAUNCH_CHECK(ctx);
  return SVK_OK;
}

extern "C" int svk_c3d2_conv42(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                               const float* d_slope, int32_t flags, float* d_out) {
  return launch_tail<Conv42>(ctx, "svk_c3d2_conv42", d_in, n_utt, d_wfrag, d_bias, d_slope, flags, d_out);
}
