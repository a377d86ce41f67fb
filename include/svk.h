/*
 * svk.h -- C-ABI of libsvk.so: the MI355X (gfx950) speaker-verification hot path.
 *
 * The reference (MingmChen/Speaker_Verification) is 100 % Python and has no FFI
 * layer; its boundary for this path is the Python API of its vendored SpeechPy
 * plus vad.py / evaluation.py / siamese.py.  Each entry point below names the
 * reference function(s) (file:line under /root/reference) whose arithmetic it
 * replaces; `speaker_verification_amd/` binds them with ctypes and re-exposes the
 * reference's own Python signatures.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer named d_* is a DEVICE pointer
 *     (hipMalloc'd / a torch tensor's data_ptr()), every h_* a host pointer.
 *   - the caller allocates every input and output; the library owns only what it
 *     returns through svk_create / svk_*_plan_create and frees it in *_destroy.
 *   - every function returns SVK_OK (0) or a negative svk_status; no exceptions,
 *     no aborts.  svk_last_error(ctx) holds a message for the last failure.
 *   - launches go to the stream set by svk_set_stream (default: the null stream)
 *     and are asynchronous w.r.t. the host; svk_sync waits for that stream.
 *   - a context is bound to one device and is not thread-safe.
 */
#ifndef SVK_H
#define SVK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVK_VERSION 111 /* 0.1.9: conv1_2's last tap as ONE [h | l] fragment (d_w2blk pair 13 = [H | H], [L | 0]: 41 MFMAs per tile, not 42), conv2_1 leaves out the column pool2 makes dead (d_act2 [..][14][32]); half-pair domain stated; 0.1.8: svk_c3d2_stage1 / svk_c3d2_stage2 / svk_c3d2_conv31 / svk_c3d2_conv32t run on the f16 matrix pipe through two-piece products (new weight tables: half-pair blocks); 0.1.7: gathered front-end input (svk_vad_energy d_src_frame -> svk_frontend_run d_src_chunk); 0.1.6: one kernel per network layer (svk_c3d2_conv32, svk_bias_prelu, svk_cube_gather_windows and the direct-form flag bits are gone); + svk_cmvn_stats, svk_cube_gather_cmvn */

typedef enum svk_status {
  SVK_OK = 0,
  SVK_ERR_BAD_ARG = -1,      /* NULL pointer, negative size, inconsistent shape   */
  SVK_ERR_UNSUPPORTED = -2,  /* a parameter combination the kernels do not cover  */
  SVK_ERR_HIP = -3,          /* a HIP runtime call failed (message has the code)  */
  SVK_ERR_NO_DEVICE = -4,    /* no usable gfx950 device                           */
  SVK_ERR_OOM = -5,
  SVK_ERR_RCCL = -6          /* RCCL could not be loaded or a collective failed   */
} svk_status;

typedef struct svk_ctx svk_ctx;
typedef struct svk_frontend_plan svk_frontend_plan;

/* ---- context, stream, memory --------------------------------------------- */
int svk_version(void);
int svk_create(int device_id, svk_ctx** out);
void svk_destroy(svk_ctx* ctx);
const char* svk_last_error(const svk_ctx* ctx);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream), NULL = null stream */
int svk_set_stream(svk_ctx* ctx, void* hip_stream);
int svk_sync(svk_ctx* ctx);
/* so that a host without torch can drive the library */
int svk_malloc(svk_ctx* ctx, size_t bytes, void** d_out);
int svk_free(svk_ctx* ctx, void* d_ptr);
int svk_memcpy_h2d(svk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int svk_memcpy_d2h(svk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int svk_memset(svk_ctx* ctx, void* d_dst, int value, size_t bytes);
/* device facts for roofline reporting: [0]=CU count, [1]=max clock kHz, [2]=LDS bytes/CU, [3]=wavefront size */
int svk_device_info(svk_ctx* ctx, int64_t out[4]);

/* ---- fused front end -------------------------------------------------------
 * PCM -> [pre-emphasis] -> frames -> rFFT power spectrum -> frame energy ->
 * mel filterbank (f32 MFMA) -> [log] -> [DCT-II (f32 MFMA)] -> [c0 := log E]
 * Replaces speechpy/feature.py:102-153 (mfcc), :156-219 (mfe), :222-258 (lmfe)
 * and, when preemph != 0, speechpy/processing.py:45-58 applied to the whole
 * clip first (circular, Q5).  Framing is the no-padding branch of
 * processing.py:112-120 (Q3) with a rectangular window (Q4).
 */
typedef enum svk_out_kind { SVK_OUT_MFE = 0, SVK_OUT_LMFE = 1, SVK_OUT_MFCC = 2 } svk_out_kind;
typedef enum svk_pcm_dtype { SVK_PCM_I16 = 0, SVK_PCM_F32 = 1 } svk_pcm_dtype;

typedef struct svk_frontend_cfg {
  int32_t frame_len;      /* samples per frame  = int(round(fs * frame_length))  */
  int32_t frame_stride;   /* samples per hop    = round(fs * frame_stride)       */
  int32_t nfft;           /* 512 or 1024 (fft_length)                            */
  int32_t num_filters;    /* 1..64                                               */
  int32_t num_ceps;       /* MFCC only: 1..num_filters                           */
  int32_t out_kind;       /* svk_out_kind                                        */
  int32_t dc_elimination; /* MFCC only: column 0 := log(frame energy) (Q8)       */
  int32_t preemph;        /* 0 = none, 1 = y[n] = x[n] - cof * x[(n-shift) mod N] */
  int32_t preemph_shift;
  float preemph_cof;
  float input_scale;      /* amplitude factor applied to the PCM first; 0 = 1.  2^-15 reads int16 PCM as
                             librosa.load hands it to lmfe (utils.py:170-173, load_data.py:50-70)          */
} svk_frontend_cfg;

/* h_filterbank: num_filters x (nfft/2+1) float64, row-major -- the matrix of
 * speechpy/feature.py:33-99, built on the host (its bin edges hinge on float64
 * libm rounding, Q1/Q2, so the host language that owns parity builds it).   */
int svk_frontend_plan_create(svk_ctx* ctx, const svk_frontend_cfg* cfg, const double* h_filterbank,
                             svk_frontend_plan** out);
void svk_frontend_plan_destroy(svk_frontend_plan* plan);
/* number of frames of a clip of n_samples: floor((n - frame_len) / stride), >= 0 (Q3) */
int64_t svk_frontend_num_frames(const svk_frontend_cfg* cfg, int64_t n_samples);
/* columns of the feature matrix: num_filters (MFE/LMFE) or num_ceps (MFCC) */
int svk_frontend_num_cols(const svk_frontend_cfg* cfg);

/* d_pcm      : concatenated clips, int16 or float32 (pcm_dtype)
 * d_offsets  : [n_utt] int64 first sample of each clip in d_pcm, or NULL = i * clip_stride
 * d_lengths  : [n_utt] int32 samples per clip, or NULL = clip_len for all
 * max_frames : row stride of the outputs (>= frames of the longest clip)
 * d_feat     : [n_utt][max_frames][num_cols] float32; rows >= n_frames are zeroed
 * d_energy   : [n_utt][max_frames] float32 frame energies (after zero handling), or NULL
 * d_n_frames : [n_utt] int32 frames produced per clip, or NULL
 * d_src_chunk: NULL, or GATHERED input (int16 PCM): clip u's signal is then the concatenation of chunks of chunk_samples
 *              samples of its PCM, chunk q = PCM chunk d_src_chunk[u * chunk_stride + q], d_lengths[u] samples of it in
 *              all (required) -- svk_vad_energy's d_src_frame with chunk_samples = frame_samples: the voiced frames of
 *              vad.py:135-168 feed the front end where they lie, without the pass that copies them to the front of a
 *              second buffer (same samples, bit-identical features).  chunk_samples a multiple of 8.
 */
int svk_frontend_run(svk_ctx* ctx, const svk_frontend_plan* plan, const void* d_pcm, int pcm_dtype,
                     const int64_t* d_offsets, const int32_t* d_lengths, int64_t clip_stride,
                     int32_t clip_len, int32_t n_utt, int32_t max_frames, float* d_feat, float* d_energy,
                     int32_t* d_n_frames, const int32_t* d_src_chunk, int32_t chunk_samples, int32_t chunk_stride);

/* ---- stage-level entry points (one speechpy function each) ----------------- */
/* processing.py:45-58.  d_in int16/float32 [n]; d_out float32 [n]; circular. */
int svk_preemphasis(svk_ctx* ctx, const void* d_in, int pcm_dtype, int64_t n, int32_t shift, float cof,
                    float* d_out);
/* processing.py:61-139.  frame t = d_sig[t*stride .. +frame_len) (zeros past n) times d_window
 * (NULL = rectangular).  d_out float32 [n_frames][frame_len].                 */
int svk_stack_frames(svk_ctx* ctx, const float* d_sig, int64_t n, int32_t frame_len, int32_t stride,
                     int32_t n_frames, const float* d_window, float* d_out);
/* processing.py:142-174.  d_frames float32 [n_frames][frame_len] -> d_out
 * [n_frames][nfft/2+1]; power = 0: |rfft| (fft_spectrum), 1: |rfft|^2/nfft (power_spectrum).
 * Frames longer than nfft are cropped, shorter are zero-padded (Q6).  Any nfft >= 2. */
int svk_spectrum(svk_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_len, int32_t nfft,
                 int32_t power, float* d_out);
/* processing.py:239-271 per clip: d_feat [n_utt][max_frames][n_cols], rows < n_frames[u]
 * (NULL = max_frames) are normalised in place; variance != 0 divides by (std + 2^-30). */
int svk_cmvn(svk_ctx* ctx, float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
             const int32_t* d_n_frames, int32_t variance);
/* The same statistics WITHOUT applying them: d_stats float64 [n_utt][2][n_cols], [u][0][c] = mean, [u][1][c] =
 * 1 / (std + 2^-30) (1 when variance == 0) over rows < n_frames[u]; clips with no rows are left untouched.  utils.py:382-397
 * (CMVN) feeds utils.py:351-379 (FeatureCube), which reads 20 x 80 rows of a clip: svk_cube_gather_cmvn below applies the
 * normalisation to just those rows on the way, instead of a pass over every row of a 145 s clip. */
int svk_cmvn_stats(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                   const int32_t* d_n_frames, int32_t variance, double* d_stats);

/* feature.py:202-217 and :146-153 on a power spectrum that is already on the device -- the general
 * (any fft_length, up to 1024 filters) counterpart of the fused front end, one workgroup per frame:
 * energy = sum of all bins (0 -> eps), mel = power x bank^T (0 -> eps), then by out_kind nothing /
 * log / log + DCT-II ortho (+ c0 := log energy).  d_bank [num_filters][n_bins] float32.
 * d_feat [n_frames][cols], cols = num_filters or num_ceps; d_energy [n_frames] or NULL. */
int svk_mel_features(svk_ctx* ctx, const float* d_power, int32_t n_frames, int32_t n_bins, const float* d_bank,
                     int32_t num_filters, int32_t out_kind, int32_t num_ceps, int32_t dc_elimination,
                     float* d_feat, float* d_energy);
/* processing.py:274-327 (cmvnw, Q10) per clip: sliding window of `win` (odd) rows, 'symmetric'
 * padding of (win-1)/2 rows; out = x - window mean; with variance != 0 a second pass divides by
 * (population std of the window over the MEAN-SUBTRACTED rows, padded the same way, + 2^-30).
 * d_in / d_out / d_tmp: [n_utt][max_frames][n_cols] float32, all distinct; d_tmp is only needed
 * (and only written) when variance != 0.  Rows >= n_frames[u] are left untouched in d_out. */
int svk_cmvnw(svk_ctx* ctx, const float* d_in, int32_t n_utt, int32_t max_frames, int32_t n_cols,
              const int32_t* d_n_frames, int32_t win, int32_t variance, float* d_tmp, float* d_out);
/* processing.py:201-236 (derivative_extraction) exactly as the reference computes it, Q11 included:
 * out[r][c] = sum_{k=1..delta} k * in[r][min(c + k, n_cols - 1)] / sum_{k} 2 k^2  (edge padding along
 * the FEATURE axis; the subtraction on processing.py:232 is a detached statement). */
int svk_derivative(svk_ctx* ctx, const float* d_in, int64_t n_rows, int32_t n_cols, int32_t delta, float* d_out);
/* processing.py:177-198 (log_power_spectrum) on a power spectrum already on the device: in place
 * p <- 10 log10(max(p, 1e-20)); normalize != 0 then subtracts the global maximum. */
int svk_log_power(svk_ctx* ctx, float* d_power, int64_t n, int32_t normalize);

/* ---- energy VAD ------------------------------------------------------------
 * vad.py:44-57 (framer, Q12) + vad.py:60-129 (ring-buffer hysteresis, Q13) with
 * the per-frame decision  sum(x^2) > threshold * frame_samples  (int64) in place of
 * webrtcvad (vad.py:90).  ring_len = int(padding_ms / frame_ms); ring_thresh =
 * floor(0.9 * ring_len) (trigger when voiced > ring_thresh, release when unvoiced >
 * ring_thresh).
 * d_keep       : [n_utt][max_vad_frames] uint8, 1 = frame is in some yielded segment
 * d_seg        : [n_utt][max_vad_frames] int32 segment ordinal or -1, or NULL
 * d_n_vad_frames: [n_utt] int32, or NULL
 * d_voiced     : int16, same offsets as d_pcm: kept frames packed to the front (the rest of a
 *                clip's slot is left untouched), or NULL
 * d_voiced_len : [n_utt] int32 samples kept (required when d_voiced or d_src_frame != NULL)
 * d_src_frame  : [n_utt][max_vad_frames] int32 or NULL: the index form of the same compaction, d_src_frame[u][q] = the
 *                frame that is the q-th kept one (entries past the kept count are left untouched); svk_frontend_run reads
 *                the PCM through it (d_src_chunk) -- with d_voiced = NULL nothing is copied
 */
int svk_vad_energy(svk_ctx* ctx, const int16_t* d_pcm, const int64_t* d_offsets, const int32_t* d_lengths,
                   int64_t clip_stride, int32_t clip_len, int32_t n_utt, int32_t frame_samples,
                   int32_t ring_len, int32_t ring_thresh, int64_t threshold, int32_t max_vad_frames,
                   uint8_t* d_keep, int32_t* d_seg, int32_t* d_n_vad_frames, int16_t* d_voiced,
                   int32_t* d_voiced_len, int32_t* d_src_frame);

/* ---- feature cube ------------------------------------------------------------
 * utils.py:351-379 (FeatureCube): out[u][0][c][r][:] = feat[u][crop[u][c] + r][:].
 * d_out float32 [n_utt][1][n_crops][crop_frames][n_cols] (a torch tensor's data_ptr()). */
int svk_cube_gather(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                    const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, float* d_out);
/* utils.py:382-397 + :351-379 in one pass: the cube of svk_cmvn-normalised features from the RAW features and svk_cmvn_stats'
 * d_stats: out = (float)(((double)feat - mean[c]) * inv[c]), the expression svk_cmvn applies (bit-identical to svk_cmvn followed
 * by svk_cube_gather); too-short clips (crop -1) give zero cubes as there. */
int svk_cube_gather_cmvn(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                         const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const double* d_stats,
                         float* d_out);

/* Crop starts drawn ON THE DEVICE (no host round trip for the per-clip frame count):
 * crop[u][c] = floor(uniform(seed, u, c) * (n_frames[u] - crop_frames)), a counter-based
 * generator (splitmix64 of seed, g, c) with g = d_utt_index[u] when that array is given (clips of a
 * ragged corpus land in batches in any order) and first_utt + u otherwise, so a draw depends only on
 * the clip's global index.  utils.py:372 draws `np.random.randint(T - 80, size=20)` from the process-global
 * NumPy RNG instead; hosts that need that exact sequence pass their own d_crop_idx to
 * svk_cube_gather.  Clips with n_frames <= crop_frames get -1 (svk_cube_gather then emits zeros)
 * and are counted in *d_bad_count (int32, may be NULL; the caller zeroes it).               */
int svk_cube_draw_crops(svk_ctx* ctx, const int32_t* d_n_frames, int32_t n_utt, int64_t first_utt,
                        const int64_t* d_utt_index, int32_t n_crops, int32_t crop_frames, uint64_t seed,
                        int32_t* d_crop_idx, int32_t* d_bad_count);

/* ---- audio ingest ----------------------------------------------------------
 * utils.py:170-173  `librosa.load(path, sr=16000, mono=True)`: down-mix (mean over
 * channels), scale int16 / 32768 and resample, for n_utt clips in one launch.
 * librosa is absent and its resampler unpinned, so the arithmetic is the published
 * polyphase windowed-sinc scheme of scipy.signal.resample_poly (zero-padded edges):
 *     y[m] = sum_j x[j] * taps[m * down + half - j * up],   half = (n_taps - 1) / 2
 * d_pcm    : [n_utt][in_stride][n_ch] interleaved int16 frames (vad.py:10-22 reads them)
 * d_in_len : [n_utt] int32 frames per clip, or NULL = clip_in
 * d_taps   : [n_taps] float32 (odd count; the host builds them, see ingest.py)
 * d_out    : [n_utt][out_stride] float32 in [-1, 1) (SVK_PCM_F32) or int16 rounded back to the
 *            16-bit grid, saturating (SVK_PCM_I16); samples past a clip's end are written as 0
 * d_out_len: [n_utt] int32 = min(ceil(n_in * up / down), clip_out), or NULL          */
int svk_ingest_resample(svk_ctx* ctx, const int16_t* d_pcm, int32_t n_ch, int64_t in_stride,
                        const int32_t* d_in_len, int32_t clip_in, int32_t n_utt, const float* d_taps,
                        int32_t n_taps, int32_t up, int32_t down, void* d_out, int32_t out_dtype,
                        int64_t out_stride, int32_t clip_out, int32_t* d_out_len);

/* ---- scoring ---------------------------------------------------------------
 * evaluation.py:67-84: cosine of every test row against every enrolled row,
 * float32, out [n_test][n_enroll]  (f32 MFMA).  dim <= 4096.                 */
int svk_cosine_scores(svk_ctx* ctx, const float* d_test, const float* d_enroll, int32_t n_test,
                      int32_t n_enroll, int32_t dim, float* d_out);
/* siamese.py:29-30: out[i] = || a[i] - b[i] ||_2 */
int svk_l2_dist(svk_ctx* ctx, const float* d_a, const float* d_b, int32_t n, int32_t dim, float* d_out);

/* ---- ROC / EER / AUC on the device ------------------------------------------------------------
 * evaluation.py:47-52 (sklearn roc_curve + roc_auc_score + brentq on interp1d) for pair sets too
 * large for the host: stable sort of the scores (descending), scan of the labels, one point per
 * distinct score, trapezoid AUC and the linear root of 1 - fpr - tpr.  d_labels: uint8, 1 = positive.
 * d_workspace: svk_roc_workspace_bytes(n) bytes.  h_out[4] (HOST) = {eer, auc, positives, ROC points};
 * the call synchronises the stream (its result is a host scalar). */
size_t svk_roc_workspace_bytes(int64_t n);
int svk_roc_eer(svk_ctx* ctx, const float* d_scores, const uint8_t* d_labels, int64_t n, void* d_workspace,
                size_t workspace_bytes, double* h_out);

/* ---- the first block of the embedding network ----------------------------------------------------
 * model.py:110-117 + :141-150 (C3D2): cube (utils.py:351-379) -> conv1_1 (1 -> 16, kernel (3,1,5)) -> BN -> PReLU
 * -> conv1_2 (16 -> 16, kernel (3,9,1), stride (1,2,1)) -> BN -> PReLU -> MaxPool3d((1,1,2)), eval mode, as ONE
 * kernel: conv1_1's output (3.3 MB per cube) lives only in LDS.  Since 0.1.8 on v_mfma_f32_16x16x32_f16 through TWO-PIECE
 * products: a value x travels as the halves h = f16(x), l = f16(x - h) (22 significant bits in four bytes) and
 * x w = h_x h_w + l_x h_w + h_x l_w, three f16 products, exact in the f32 they are accumulated in (measured on this network:
 * 1 - 5e-7 of the activation scale from the f64 convolution; the f32 direct form: 2 - 8e-7).  Direct form (the depth
 * transform's adds do not distribute over pieces), two taps per K = 32 block.  The host folds the BatchNorm statistics into
 * weights / biases, splits the weights and lays them out in the lane order of the MFMA's A operand (lane l = (co = l & 15,
 * kk = l >> 4), eight halves: K = 8 kk + e):
 *   d_w1blk [2][64][8 halves]: conv1_1, element e = tap t = 8 (kk & 1) + e (t = 5 kd + kw; t = 15 -> 0):
 *                        block 0 = H for every kk, block 1 = L for kk < 2 and 0 for kk >= 2;   d_bias1 [16]
 *   d_w2blk [14][2][64][8 halves]: conv1_2's tap pairs (a | b): pr < 12 -> a = (kd = pr / 4, kh = 2 (pr % 4)), b = (kd, kh + 1);
 *                        pr = 12 -> (0, 8) | (1, 8).  Element e = W2[co][ci = 8 (kk & 1) + e][tap a if kk < 2 else b];
 *                        block 0 = H pieces, block 1 = L pieces.  pr = 13 is the last tap (2, 8) ALONE, which the kernel reads
 *                        as one [h | l] fragment: block 0 = its H pieces at every kk, block 1 = its L pieces for kk < 2, 0 above;
 *                        d_bias2 [16];   d_slope1 / d_slope2 [16] PReLU slopes per channel
 * DOMAIN of the half-pair kernels (this one, svk_c3d2_stage2 / conv31 / conv32t / conv41): a value carries 22 significant bits
 * while |x| >= 2^-3, an absolute error floor of 2^-25 below that, and must stay under 65 504 (an f16's largest finite value; the
 * reference's features -- log mel energies, MFCCs, CMVN output -- are within +-100).  Weights have the same floor, which is why
 * the host fixes every channel's scale before it splits them (model.FusedEmbedder: activations of channel c are carried times
 * the power of two nearest 1 / (|gamma_c| + |beta_c|) of its BatchNorm, the next layer's weights take the inverse: exact, and
 * the tables no longer depend on how a checkpoint distributes a channel's scale between one layer and the next).
 * d_feat [n_utt][max_frames][40] f32, d_crop_idx [n_utt][20] as for svk_cube_gather (a start outside the clip -> zero rows).
 * d_out: the activation after the pool, float32, channels last: [n_utt][16 d][36 h][18 w][16 c]
 * flags bit 1 (value 2) = the caller asserts every PReLU slope lies in [0, 1] (then prelu(v) = max(v, slope v): two
 * instructions per value instead of four); every other bit must be 0.
 * Geometry other than the 20 x 80 x 40 cube -> SVK_ERR_UNSUPPORTED (the torch module is the path for other models). */
size_t svk_c3d2_stage1_lds_bytes(void);
int svk_c3d2_stage1(svk_ctx* ctx, const float* d_feat, int32_t n_utt, int32_t max_frames, int32_t n_cols,
                    const int32_t* d_crop_idx, int32_t n_crops, int32_t crop_frames, const void* d_w1blk,
                    const float* d_bias1, const float* d_slope1, const void* d_w2blk, const float* d_bias2,
                    const float* d_slope2, int32_t flags, float* d_out);

/* The second block, model.py:119-124 + :151-158: conv2_1 (16 -> 32, kernel (3,1,4)) -> BN -> PReLU -> conv2_2
 * (32 -> 32, kernel (3,8,1), stride (1,2,1)) -> BN -> PReLU -> MaxPool3d((1,1,2)), two kernels on v_mfma_f32_16x16x32_f16
 * through two-piece products like svk_c3d2_stage1 (direct form; the input region of a work item is split into (h, l) halves
 * while it is staged into LDS, the weights of all taps sit in registers); epilogues carry bias, PReLU and the pool.
 *   d_in     [n_utt][16][36][18][16]  = svk_c3d2_stage1's output
 *   d_w21blk [2 nt][6 pairs][2][64][8 halves]: conv2_1, lane l = (co = 16 nt + (l & 15), kk = l >> 4), element e =
 *            W[co][ci = 8 (kk & 1) + e][kd][kw + (kk >= 2)] of the tap pair 2 kd + kw / 2 (kw = 0, 2); block 0 = H, 1 = L
 *   d_w22blk [2 nt][24 taps][2][64][8 halves]: conv2_2, element e = W[co][ci = 8 kk + e][kd][kh], tap = 8 kd + kh; H | L
 *   d_bias / d_slope [32] per layer (BN folded; PReLU slope per channel)
 *   d_act2   [n_utt][14][36][14][32]  conv2_1's activation (scratch, f32): 14 of the layer's 15 columns -- conv2_2 is one column
 *            wide and pool2 drops its 15th, so conv2_1's 15th is never read and (since 0.1.9) never computed;
 *   d_out    [n_utt][12][15][7][32] (channels last)
 * flags as for svk_c3d2_stage1. */
int svk_c3d2_stage2(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_w21blk, const float* d_bias21,
                    const float* d_slope21, const void* d_w22blk, const float* d_bias22, const float* d_slope22,
                    int32_t flags, float* d_act2, float* d_out);

/* conv3_1 (32 -> 64, kernel (3,1,3)) -> BN -> PReLU, model.py:126-128 + :159-161, one kernel on v_mfma_f32_16x16x32_f16
 * through two-piece products like svk_c3d2_stage1 (direct form, one tap per K = 32 block).
 *   d_in    [n_utt][12][15][7][32]   = svk_c3d2_stage2's output
 *   d_wblk  [4 nt][9 taps][2][64][8 halves]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: W31[co][ci = 8 kk + e][kd][kw],
 *           tap 3 kd + kw (BatchNorm folded); block 0 = H = f16(w), block 1 = L = f16(w - H);  d_bias / d_slope [64]
 *   flags   bit 1: the caller asserts every PReLU slope lies in [0, 1]; every other bit must be 0
 *   d_out   [n_utt][10 d][8 chunks of 8 channels][5 w][15 h][8]: chunked and column-major, what svk_c3d2_conv32t stages from */
int svk_c3d2_conv31(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                    const float* d_slope, int32_t flags, float* d_out);

/* conv3_2 (64 -> 64, kernel (3,7,1)) -> BN -> PReLU, model.py:129-131 + :162-164, one kernel on v_mfma_f32_16x16x32_f16
 * through two-piece products like svk_c3d2_stage1 (direct form; per (cube, column) work items; csrc/c3d2.hip):
 *   d_in    [n_utt][10][8][5][15][8] = svk_c3d2_conv31's output
 *   d_wblk  [4 nt][2 kb][21 taps][2][64][8 halves]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: W32[co][ci = 32 kb + 8 kk + e][kd][kh],
 *           tap 7 kd + kh (BatchNorm folded); block 0 = H = f16(w), block 1 = L = f16(w - H);  d_bias / d_slope [64]
 *   d_out   [n_utt][8][8][45][8]     = what svk_c3d2_conv41 takes                                                       */
int svk_c3d2_conv32t(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                     const float* d_slope, int32_t flags, float* d_out);
/* conv4_1 (64 -> 128, kernel (3,1,3)) -> BN -> PReLU, model.py:132-135 + :165-166, one kernel on v_mfma_f32_16x16x32_f16
 * through two-piece products like svk_c3d2_stage1 (direct form; one cube per work item, staged whole; csrc/c3d2.hip):
 *   d_in    [n_utt][8][8][45][8]     = svk_c3d2_conv32t's output
 *   d_wblk  [8 nt][9 taps][2 kb][2][64][8 halves]: lane (co = 16 nt + (l & 15), kk = l >> 4), e: W41[co][ci = 32 kb + 8 kk + e][kd][kw],
 *           tap 3 kd + kw (BatchNorm folded); block 0 = H = f16(w), block 1 = L = f16(w - H);  d_bias / d_slope [128]
 *   flags   bit 1: the caller asserts every PReLU slope lies in [0, 1]; every other bit must be 0
 *   d_out   [n_utt][6][16][27 = 9 h x 3 w][8]: chunked [cube][depth][channel / 8][pixel][channel % 8], what svk_c3d2_conv42 takes */
int svk_c3d2_conv41(svk_ctx* ctx, const float* d_in, int32_t n_utt, const void* d_wblk, const float* d_bias,
                    const float* d_slope, int32_t flags, float* d_out);
/* The end of the network, model.py:136-139 (definitions) + :167-170 (forward): conv4_2 (128 -> 128, kernel (3,7,1)) -> BN -> PReLU,
 * flatten, FC5 (4 608 -> 128) -- GEMMs over the BATCH on v_mfma_f32_16x16x4_f32 (csrc/c3d2_tail.hip): an M tile is one output
 * position of 16 cubes; conv4_2 runs through Winograd's F(2, 3) along depth with the input transform applied once while a
 * chunk is staged into LDS and the weight transform applied by the HOST.
 *   svk_c3d2_conv42  d_in  [n_utt][6][16][27][8] = svk_c3d2_conv41's output
 *                    d_wfrag [8 nt][16 chunks][7 kh][4 k][64][2]: lane (co = 16 nt + (l & 15), kk = l >> 4), e:
 *                            G_k[co][8 chunk + 2 kk + e][kh], G0 = g0, G1 = (g0 + g1 + g2) / 2, G2 = (g0 - g1 + g2) / 2,
 *                            G3 = g2 over the three depth taps g of the BN-folded weights;  d_bias / d_slope [128]
 *                    d_out [n_utt][4][16][9 = 3 h x 3 w][8]
 *   flags            bit 1: the caller asserts every PReLU slope lies in [0, 1]
 *   svk_c3d2_fc5     d_in  [n_utt][4 608] = svk_c3d2_conv42's output, K index ((d * 16 + chunk) * 9 + pixel) * 8 + channel % 8
 *                    d_wfrag [4 d][8 nt][72][64][4]: lane (j = 16 nt + (l & 15), kk = l >> 4), e: W5[j][column of K index
 *                            1 152 d + 16 step + 4 kk + e] (model.py:168 flattens NCDHW: column = channel * 36 + d * 9 + pixel)
 *                    d_bias [128];  d_work: svk_c3d2_fc5_workspace_floats(n_utt) floats (partial sums of the four K ranges,
 *                    added in a fixed order: bitwise repeatable);  d_out [n_utt][128]                                  */
int svk_c3d2_conv42(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias,
                    const float* d_slope, int32_t flags, float* d_out);
size_t svk_c3d2_fc5_workspace_floats(int32_t n_utt);
int svk_c3d2_fc5(svk_ctx* ctx, const float* d_in, int32_t n_utt, const float* d_wfrag, const float* d_bias, float* d_work,
                 float* d_out);


/* ---- multi-GPU: the one exchange step of the path ------------------------------------------------
 * Utterances shard over the GPUs of a node with no data-path exchange until scoring; then every rank needs
 * the enrolled embeddings: ONE all-gather of the [rows_per_rank][dim] float32 shards over RCCL / xGMI
 * (the reference has no collective, only in-process DataParallel, train.py:40-41).  One process
 * per GPU, one context per process.  A torch host uses torch.distributed instead (distributed.py); these
 * entry points serve a host without torch.  RCCL is loaded at the first call (dlopen), not at link time.
 *   rank 0: svk_comm_unique_id(ctx, id)  -> ship the 128 bytes to the other ranks by any host channel
 *   all   : svk_comm_init(ctx, id, n_ranks, rank)            (collective: every rank must call it)
 *   all   : svk_allgather_f32(ctx, d_send, d_recv, count)    d_recv holds n_ranks x count floats, rank-major;
 *           asynchronous on the context's stream like every launch
 *   all   : svk_comm_destroy(ctx)                                                                          */
int svk_comm_unique_id(svk_ctx* ctx, char out[128]);
int svk_comm_init(svk_ctx* ctx, const char id[128], int32_t n_ranks, int32_t rank);
int svk_allgather_f32(svk_ctx* ctx, const float* d_send, float* d_recv, size_t count_per_rank);
int svk_comm_destroy(svk_ctx* ctx);
/* out[0] = ranks of the context's communicator (0 = none), out[1] = this rank */
int svk_comm_info(const svk_ctx* ctx, int32_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* SVK_H */
