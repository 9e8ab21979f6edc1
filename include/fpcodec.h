/*
 * fpcodec.h -- C ABI of libfpcodec.so (MI355X / gfx950).
 *
 * Drop-in boundary for the inference hot path of
 * haiciyang/Feature-predictor-for-speech-codec.  The reference has no FFI of its
 * own (pure Python); every entry point below names the reference interface it
 * replaces (file:line under /root/reference) and is what a ctypes binding on the
 * reference side calls.
 *
 * Conventions
 *  - every function returns 0 (FPC_OK) or a negative fpc_status; nothing throws
 *  - fpc_last_error() returns the text of the last failure on the calling thread
 *  - "dev" pointers are device (HIP) pointers, "host" pointers are host memory
 *  - handles are bound to the device that was current at creation
 *  - launches are asynchronous on the given hipStream_t (passed as void*;
 *    NULL = the null stream); no hidden synchronisation in the *_run calls
 *  - there is NO CPU fallback: without a HIP device every create call fails
 */
#ifndef FPCODEC_H
#define FPCODEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fpc_encode takes the input mask (mask_dev, NULL = thresholds); exports fpc_build_info, fpc_lpcnet_set_chunk_frames
 *    (fpc_lpcnet_workspace_bytes follows the handle's chunk setting).  A binding compares fpc_abi_version() with the
 *    FPC_ABI_VERSION it was written against BEFORE any other call: the symbol sets of versions differ.
 * 3: exports fpc_kmeans1d (scalar-codebook k-means) and fpc_predictor_fallback_groups; fpc_predictor_forward keeps a scratch
 *    block in the handle (relu(h2) of all frames for the batched output layer).
 * 4: exports fpc_lpcnet_set_pairing and fpc_lpcnet_last_streams_per_workgroup (fpc_lpcnet_synthesize decodes two utterances
 *    per workgroup when the batch exceeds the device's compute units), fpc_lpcnet_paired_utterances and fpc_selftest. */
#define FPC_ABI_VERSION 4
#define FPC_API __attribute__((visibility("default")))

typedef enum {
    FPC_OK = 0,
    FPC_ERR_INVALID = -1, /* bad argument / shape */
    FPC_ERR_HIP = -2,     /* HIP runtime error (text in fpc_last_error) */
    FPC_ERR_NO_DEVICE = -3,
    FPC_ERR_CAPACITY = -4, /* model does not fit the on-chip layout */
    FPC_ERR_TIMEOUT = -5,  /* a row-split exchange of a predictor launch gave up (see fpc_predictor_status) */
    FPC_ERR_NONFINITE = -6 /* a NaN / infinite residual reached a quantizer (see fpc_predictor_status) */
} fpc_status;

typedef void* fpc_stream; /* hipStream_t */

FPC_API const char* fpc_last_error(void);
FPC_API int fpc_abi_version(void);
/* "fpcodec abi <n> gfx950" followed by every compile-time tunable (-DFPC_...) that differs from the shipped default and
 * every diagnostic switch the library was built with: a non-default build identifies itself (the reference has no
 * counterpart: its scripts are the build). */
FPC_API const char* fpc_build_info(void);
/* number of visible HIP devices (0 on a CPU-only host); never fails */
FPC_API int fpc_device_count(void);
/* Self-check of the library's device-buffer bookkeeping, runnable with or without a GPU: a failed allocation (no device, or an
 * absurd size on a device) must leave a buffer empty -- pointer NULL, size 0 -- so that a later "large enough?" test cannot
 * pass on a block that is not there (a forward call would otherwise launch on a null scratch block).  0 when the invariants
 * hold, a negative fpc_status with text in fpc_last_error otherwise.  (No reference counterpart.) */
FPC_API int fpc_selftest(void);

/* ------------------------------------------------------------------------
 * Feature predictor  (src/models/wavernn.py:24-52 parameters,
 *                     :63-102 forward, :165-256 encoder)
 * Weight pointers are HOST pointers in the PyTorch state_dict layout
 * (row-major, GRU gate rows ordered [r; z; n]).
 * ---------------------------------------------------------------------- */
typedef struct {
    int in_features; /* 20 */
    int gru_units1;  /* 384 */
    int gru_units2;  /* 128 */
    int fc_units;    /* 18  */
    const float* rnn1_weight_ih; /* [3*H1, in]  rnn1.weight_ih_l0 */
    const float* rnn1_weight_hh; /* [3*H1, H1]  rnn1.weight_hh_l0 */
    const float* rnn1_bias_ih;   /* [3*H1] */
    const float* rnn1_bias_hh;   /* [3*H1] */
    const float* rnn2_weight_ih; /* [3*H2, H1]  rnn2.weight_ih_l0 */
    const float* rnn2_weight_hh; /* [3*H2, H2] */
    const float* rnn2_bias_ih;   /* [3*H2] */
    const float* rnn2_bias_hh;   /* [3*H2] */
    const float* fc_weight;      /* [fc, H2]    dual_fc.0.weight */
    const float* fc_bias;        /* [fc]        dual_fc.0.bias */
} fpc_predictor_weights;

typedef struct fpc_predictor fpc_predictor;

FPC_API int fpc_predictor_create(const fpc_predictor_weights* w, fpc_predictor** out);
/* drops the caller's reference; a live fpc_trainer built on the handle keeps the device weights alive */
FPC_API void fpc_predictor_destroy(fpc_predictor* p);

/* Kernel forms of the predictor.  All give the same bits (every row is evaluated in the canonical order of
 * oracle/fpc_oracle.c, matvec_seg / matvec_t); the tests compare them with each other and with the oracle.
 *
 * (1) Weights-stationary (csrc/predictor_ws.h, predictor_wsd.h, predictor_bwd_ws.h) -- what fpc_predictor_forward,
 *     fpc_encode, fpc_decode_features and fpc_trainer_step (forward AND backward pass) run for the reference's production
 *     shape (20 -> 384 -> 128 -> 18) on a whole MI355X (>= 256 CUs) unless a row split is pinned (below) or the codebooks
 *     exceed the limits of its search (more than 1 024 entries in a stage, more than 256 codes in a scalar book).
 *     Utterances are taken in GROUPS of 16 (= the M dimension of one f32 MFMA tile); a group runs on 32 workgroups, which
 *     the launch arranges to be the 32 CUs of one XCD (8 groups = 128 utterances fill the chip; more groups follow as
 *     workgroups retire; a partly filled last group costs what a full one does).  Each workgroup keeps 1/32 of every
 *     weight matrix on chip for the whole launch; the new state values (backward: the gate gradients) go round as 16-byte
 *     granules {epoch, 3 values} in a block that belongs to the predictor handle and is cleared on the stream before each
 *     launch, two granule sets per hop used by frame parity.
 * (2) Row split (csrc/predictor.hip, predictor_df.h) -- the generic-shape family: every other shape, FPC_PRED_WS=0, a
 *     pinned split, and the FALLBACK of (1): one utterance on 1, 2, 4 or 8 workgroups (one CU each) while the batch leaves
 *     CUs idle (B x n <= number of CUs), the slices of the recurrent state exchanged as tagged 8-byte words; weights
 *     streamed from L2 every frame.  Two-role kernels: three waves walk a frame's latency chain, the others stream the
 *     recurrent products; LDS counters instead of workgroup barriers.
 * Residency -- what (1) needs and what happens without it.  The frame loop of (1) needs all 32 workgroups of a group
 *     RESIDENT at once (each spins for the others' granules; one workgroup per CU: ~150 kB of LDS), and nothing guarantees
 *     that: another process, or this library's own vocoder launch on a side stream, may hold CUs for as long as it runs.
 *     So every group DECIDES once, before anything is computed or written: each workgroup publishes a hello granule and
 *     waits at most 10 ms for the other 31; the first to see them all proposes GO, the first to lose patience FALLBACK (one
 *     compare-and-swap on a word of the granule block, the first proposal wins, every workgroup of the group -- also those
 *     dispatched later -- adopts it).  GO: all 32 exist and stay resident; the launch runs as described.  FALLBACK: every
 *     workgroup of the group returns at once, and the row-split launch (2) that every entry point queues BEHIND its
 *     weights-stationary launch on the same stream -- one workgroup per utterance, no partner to wait for -- serves exactly
 *     the groups that decided so (normally none: B workgroups that read one word and return, ~4 us).  Slower, never
 *     wrong, never a timeout: fpc_encode beside a long fpc_lpcnet_synthesize on another stream returns the quiet run's
 *     bits (tests).  fpc_predictor_fallback_groups() reports how many groups of the last launch took that route.
 *     Row split with n > 1 (automatic only for shapes / switches that select (2) as the first choice) has no such
 *     decision: it assumes the process owns the GPU; shared-GPU deployments of those pin one workgroup per utterance
 *     with fpc_predictor_set_split(p, 1) (or FPC_PRED_SPLIT=0), as bench.py does when ranks share a device.
 * Common to (1) and (2):
 *  - Placement is arranged for, checked, never assumed.  Blocks b and b + 8 are observed to land on one XCD (round-robin
 *    dealing), so the workgroups of a group get block indices 8 apart; each reads its XCD id from the hardware register
 *    (s_getreg_b32 HW_REG_XCC_ID), the ids go round with the hello granules through the general path, and only if ALL
 *    agree the exchange uses plain stores, which stay in that XCD's L2 where the partners' L1-bypassing (sc1) loads find
 *    them (measured: tools/ubench/ub5.hip -- plain stores are seen by sc1 loads inside an XCD, never across XCDs).
 *    Otherwise -- and always with FPC_FAST_HOP=0, which the tests use to run this path on a one-XCD placement -- every
 *    exchanged value is written through (sc1 stores).  A different dealing, or a wrong XCD id, therefore costs speed,
 *    never correctness: the general path assumes nothing about where a workgroup runs.
 *  - Once a group runs, a spin that does not see its partner within 1 s of wall clock gives up, never hangs (a partner
 *    that WAS there and stopped answering: a real failure, not a placement question): the launch stores NaN (floats) and
 *    -2 (symbols) for the utterance (row split: from that frame on; weights-stationary: for the whole group from the
 *    frame before on), counts those frames in no histogram, the training step skips its Adam update, and the handle's
 *    sticky status word turns every later call on the handle -- and, at once, every call that synchronises anyway
 *    (fpc_decode_features, fpc_trainer_step with loss_host, fpc_trainer_export) -- into FPC_ERR_TIMEOUT with text in
 *    fpc_last_error(), until fpc_predictor_status() has reported and cleared it.  The asynchronous entry points
 *    (fpc_predictor_forward, fpc_encode) therefore return FPC_OK for the failing launch itself: a caller that consumes
 *    their outputs without another call on the handle asks fpc_predictor_status() first.
 *  - Launches of ONE handle may be issued on different streams: a call on another stream first waits (on the device)
 *    for the handle's previous launch.  Creating and destroying handles is thread-safe; calls on one handle are not.
 * Environment: FPC_PRED_WS=0 never the weights-stationary kernels; FPC_PRED_SPLIT=0 one workgroup per utterance, 2|4|8
 * exactly that many (either selects the row-split kernels); FPC_TRAIN_BWD_ROWSPLIT=1 the training step's backward pass on
 * the row-split kernel; FPC_FAST_HOP=0 the write-through exchange everywhere.  Test hooks: FPC_SPIN_LIMIT_US /
 * FPC_HELLO_LIMIT_US (shorter bounds of the frame loop's wait / of the residency decision), FPC_TEST_WITHHOLD_PUBLISH=1
 * (the last workgroup of utterance 0 / group 0 never publishes a frame's values: the give-up path), =backward (the same in
 * the training step's backward kernel alone) or =hello (not even its hello: stands for a workgroup that is not resident: the
 * fallback path). */

/* Synchronises the device and returns what the launches on the handle have reported: FPC_OK, FPC_ERR_TIMEOUT (a
 * row-split exchange gave up) or FPC_ERR_NONFINITE (a NaN / infinite residual reached a quantizer in fpc_encode: those
 * frames carry the symbols -2 and were not searched); clears the condition. */
FPC_API int fpc_predictor_status(fpc_predictor* p);
/* Diagnostic: how many groups of 16 utterances of the handle's LAST weights-stationary launch could not get their 32
 * workgroups resident within the bound and were served by the row-split launch behind it ("Kernel forms" above: correct
 * either way, slower).  Synchronises the device; >= 0, or a negative fpc_status.  (No reference counterpart.) */
FPC_API int fpc_predictor_fallback_groups(fpc_predictor* p);
/* workgroups per utterance: 0 automatic (default), 1 never split, 2 / 4 / 8 exactly that many when the shape allows */
FPC_API int fpc_predictor_set_split(fpc_predictor* p, int n);

/* Wavernn.forward (wavernn.py:63-102): x [B,L,in] -> y [B,L,fc]; h1 [B,H1],
 * h2 [B,H2] are read as initial state and overwritten with the final state.
 * All pointers are device pointers.  (Production shape: the recurrences run in k_forward_ws, which keeps relu(h2) of every
 * frame in a scratch block of the handle -- B * L * H2 floats, grown on demand -- and the output layer runs over all frames
 * at once behind it, k_out_layer: same values, same order, y bit-identical to the layer evaluated frame by frame.) */
FPC_API int fpc_predictor_forward(fpc_predictor* p, const float* x_dev, int B, int L,
                          float* h1_dev, float* h2_dev, float* y_dev, fpc_stream s);

/* Codebooks (src/quantization/vq_func.py:134-185; file formats written by
 * src/train_cb.py:125-130,217-221).  HOST pointers, float64.
 *   vq_hi : S_hi stages (1 or 2), stage s has N_hi[s] rows of 17 doubles, stages
 *           stored back to back                      (cfg['cb_path'])
 *   vq_lo : one stage, N_lo rows of 17 (may be NULL/0) (cfg['bl_cb_path'])
 *   scl_hi: n_hi scalars                              (cfg['scl_cb_path'])
 *   scl_lo: n_lo scalars (may be NULL/0)              (cfg['bl_scl_cb_path'])
 */
typedef struct fpc_codebooks fpc_codebooks;
FPC_API int fpc_codebooks_create(const double* vq_hi, int S_hi, const int* N_hi,
                         const double* vq_lo, int N_lo,
                         const double* scl_hi, int n_hi,
                         const double* scl_lo, int n_lo, fpc_codebooks** out);
FPC_API void fpc_codebooks_destroy(fpc_codebooks* c);

/* Histogram block layout of fpc_encode's `hist` (= cb_tot of wavernn.py:189):
 * [n_hi | n_lo | N_hi[0] | N_hi[1] | N_lo] unsigned 64-bit counters, absent
 * codebooks contribute zero-length segments. */
FPC_API int fpc_codebooks_hist_size(const fpc_codebooks* c);

/* Wavernn.encoder (wavernn.py:165-256).  Device pointers.
 *   feat   [B,L,20]   normalised features (cepstrum/24.1, pitch/24.1)
 *   c_in   [B,L,20]   = reference c_in[:,1:,:]
 *   r, r_qtz, r_under [B,L,18]
 *   ind1, ind2 [B,L]  float 0/1 (reference ind1_mask/ind2_mask)
 *   idx    [B,L,4]    int32 {scalar idx, vq stage-1, vq stage-2, below-thr vq idx};
 *                     -1 where not coded; scalar idx is offset by +n_hi when it
 *                     came from the below-threshold scalar codebook. May be NULL.
 *   hist   see above; ADDED to (caller zeroes). May be NULL.
 *   mask   [B,L,2]    float, or NULL.  NULL: the thresholds l1 / l2 decide per frame what is coded with the
 *                     above-threshold books (wavernn.py:201-207).  Not NULL: the input-mask mode (wavernn.py:209-211) --
 *                     mask[b,i,0] != 0 selects the above-threshold scalar book for c0, mask[b,i,1] != 0 the
 *                     above-threshold VQ for c1..c17 (`if ind1[k,0]`, :218), l1 / l2 are ignored, ind1 / ind2 are
 *                     written as zeros (the reference fills them from the thresholds only) and, with qtz=0, r and
 *                     r_under are the products with the mask's own values (r_s * mask, r_s * (1 - mask), :245-249).
 * qtz=0 reproduces the un-quantised branch (wavernn.py:244-252); cb may then be NULL. */
FPC_API int fpc_encode(fpc_predictor* p, const fpc_codebooks* cb, const float* feat_dev, int B, int L,
               float l1, float l2, int qtz, float* c_in_dev, float* r_dev, float* r_qtz_dev,
               float* r_under_dev, float* ind1_dev, float* ind2_dev, int32_t* idx_dev,
               unsigned long long* hist_dev, const float* mask_dev, fpc_stream s);

/* Receiver side of fpc_encode (the reference's own Wavernn.decoder, wavernn.py:367-379,
 * is dead code): rebuilds c_in [B,L,20] from the symbols alone -- idx [B,L,4] exactly as fpc_encode wrote
 * them, pitch [B,L,2] = feat[:,:,18:20] (side information) -- with the same predictor steps and the same
 * dequantisation, so the result equals the encoder's c_in bit for bit.  Synchronises the stream; fails if
 * a symbol lies outside its codebook. */
FPC_API int fpc_decode_features(fpc_predictor* p, const fpc_codebooks* cb, const float* pitch_dev,
                        const int32_t* idx_dev, int B, int L, float* c_out_dev, fpc_stream s);

/* Stand-alone quantizers with the reference call shapes
 * (vq_quantize vq_func.py:134, scl_quantize vq_func.py:167).  Device pointers.
 *   which = 0: above-threshold codebook, 1: below-threshold codebook
 *   r [n,17] float32 -> qr [n,17] float64, idx [n,2] int32 (stage 2 = -1 if absent) */
FPC_API int fpc_vq_quantize(const fpc_codebooks* cb, int which, const float* r_dev, int n,
                    double* qr_dev, int32_t* idx_dev, fpc_stream s);
/*   x [n] float32 -> q [n] float64, idx [n] int32 */
FPC_API int fpc_scl_quantize(const fpc_codebooks* cb, int which, const float* x_dev, int n,
                     double* q_dev, int32_t* idx_dev, fpc_stream s);

/* ceps2lpc_v (src/ceps2lpc/ceps2lpc_vct.py:122-162): ceps [N,stride] float32
 * (first 18 columns used, un-normalised i.e. already x24.1) -> lpc [N,16].
 * Optional outputs (may be NULL): e [N] final Levinson error, rc [N,16] reflection
 * coefficients (the reference returns both for the LAST row only). */
FPC_API int fpc_ceps2lpc(const float* ceps_dev, int N, int stride, float* lpc_dev, float* e_dev,
                 float* rc_dev, fpc_stream s);

/* ------------------------------------------------------------------------
 * LPCNet-style vocoder (NOT in /root/reference: xiph/LPCNet training_tf2/
 * lpcnet.py + test_lpcnet.py, call site README.md:47; spec in DESIGN.md).
 * Weight pointers are HOST pointers in Keras layouts.
 * ---------------------------------------------------------------------- */
typedef struct {
    const float* embed_pitch;     /* [256,64]            */
    const float* conv1_kernel;    /* [3,84,128]  (tap,in,out) */
    const float* conv1_bias;      /* [128] */
    const float* conv2_kernel;    /* [3,128,128] */
    const float* conv2_bias;      /* [128] */
    const float* dense1_kernel;   /* [128,128] (in,out) */
    const float* dense1_bias;     /* [128] */
    const float* dense2_kernel;   /* [128,128] */
    const float* dense2_bias;     /* [128] */
    const float* embed_sig;       /* [256,128] */
    const float* gru_a_kernel;    /* [512,1152] rows: sig|pred|exc|cfeat, cols z|r|h */
    const float* gru_a_recurrent; /* [384,1152] dense storage, block-sparse content */
    const float* gru_a_bias;      /* [2,1152]   (input bias ; recurrent bias) */
    const float* gru_b_kernel;    /* [512,48]   rows: gru_a out(384)|cfeat(128) */
    const float* gru_b_recurrent; /* [16,48] */
    const float* gru_b_bias;      /* [2,48] */
    const float* md_kernel;       /* [256,16,2] MDense kernel (unit,in,channel) */
    const float* md_bias;         /* [256,2] */
    const float* md_factor;       /* [256,2] */
} fpc_lpcnet_weights;

typedef struct fpc_lpcnet fpc_lpcnet;

FPC_API int fpc_lpcnet_create(const fpc_lpcnet_weights* w, fpc_lpcnet** out);
FPC_API void fpc_lpcnet_destroy(fpc_lpcnet* m);

/* bytes of device workspace fpc_lpcnet_synthesize / fpc_lpcnet_condition need for (B,T) with the handle's current
 * chunk setting: B*T*1584 floats for a whole-utterance pass, B*(chunk*1584 + 768) floats -- independent of T -- for a
 * chunked one */
FPC_API long long fpc_lpcnet_workspace_bytes(const fpc_lpcnet* m, int B, int T);

/* Frames per pass of fpc_lpcnet_synthesize (0, the default: the whole utterance in one pass; the environment variable
 * FPC_LPCNET_CHUNK sets the default of new handles).  With frames > 0 an utterance longer than that is synthesised chunk by
 * chunk -- frame-rate layers and conditioning products of a chunk (the convolutions' halo frames evaluated again), then the
 * sample loop over it, the per-stream state (both GRU states, history ring, control block, de-emphasis) carried in a
 * 2 kB record per utterance -- so the workspace does not grow with T.  Same samples bit for bit (tests); each chunk ends
 * when its slowest utterance does, which costs time: the default stays one pass. */
FPC_API int fpc_lpcnet_set_chunk_frames(fpc_lpcnet* m, int frames);

/* How fpc_lpcnet_synthesize maps utterances onto workgroups.  mode 0 (default; the environment variable
 * FPC_LPCNET_PAIRING sets the default of new handles): one utterance per workgroup (k_decode) while B <= the device's
 * compute units; for larger batches -- the grid of B workgroups would run in rounds -- the first part of the batch goes two
 * utterances per workgroup (k_decode2: both walk the sample loop in lockstep and share the weights in registers / LDS, every
 * barrier and every L2 round trip; a round of 2 x CUs utterances costs about 1.56 rounds of k_decode) and the rest, if that
 * is cheaper, as one more round of k_decode: all paired for CUs < B <= 2 CUs, one paired round + one plain round for
 * 2 CUs < B <= 3 CUs, and so on.  mode 1: all utterances paired (B >= 2); mode -1: never.  The PCM of an utterance does not
 * depend on the mapping (same operations in the same order; tests).  Models whose update / reset row groups are wider than
 * 4 lanes (fpc_lpcnet_kernel_variant 1616), or whose row groups do not fit the mat-vec lanes once padded to even widths,
 * always take one utterance per workgroup. */
FPC_API int fpc_lpcnet_set_pairing(fpc_lpcnet* m, int mode);
/* The split mode 0 makes (a pure function, no device needed): of B utterances on a device of `cus` compute units, how many --
 * the first ones of the batch -- go two per workgroup; the rest go one per workgroup in one more round.  0 for B <= cus. */
FPC_API int fpc_lpcnet_paired_utterances(int B, int cus);
/* utterances per workgroup of the last fpc_lpcnet_synthesize on this handle: 1 or 2 (0 before the first call, <0 on a
 * null handle) */
FPC_API int fpc_lpcnet_last_streams_per_workgroup(const fpc_lpcnet* m);

/* test_lpcnet.py loop.  Device pointers.
 *   features [B,T,36] float32 (un-normalised: cepstrum, pitch, corr, 16 LPC)
 *   seeds    [B]      uint64  (Philox key of each utterance)
 *   pcm      [B,T*160] int16  de-emphasised output; the first 17 samples of each
 *                             utterance are 0 (test_lpcnet.py skips order+1)
 *   workspace: at least fpc_lpcnet_workspace_bytes(B,T) bytes */
FPC_API int fpc_lpcnet_synthesize(fpc_lpcnet* m, const float* features_dev, int B, int T,
                          const uint64_t* seeds_dev, int16_t* pcm_dev, void* workspace_dev,
                          fpc_stream s);

/* frame-rate conditioning only (enc of lpcnet.py): cfeat [B,T,128] */
FPC_API int fpc_lpcnet_condition(fpc_lpcnet* m, const float* features_dev, int B, int T,
                         float* cfeat_dev, void* workspace_dev, fpc_stream s);

/* duration (ms) of the last call's decode-kernel launch measured with HIP events on the launch stream (a chunked
 * pass: from its first sample loop to its last, the chunks' frame-rate launches in between included); synchronises
 * on those events. <0 if none. */
FPC_API float fpc_lpcnet_last_decode_ms(fpc_lpcnet* m);

/* diagnostic: which decode-kernel instance the model's sparsity pattern selects, as
 * 100 * (partial-sum planes of the update/reset gates) + (planes of the candidate gate):
 * 208, 408 or 1616 ("sparse product").  <0 on a null handle. */
FPC_API int fpc_lpcnet_kernel_variant(const fpc_lpcnet* m);

/* ---- predictor training step (src/train_frame.py:53-120, the live branch) ----
 * One step = teacher-forced Wavernn.forward over feat [B,L,in] (device), nn.MSELoss(out[:, :-1],
 * feat[:, 1:, :fc]), back-propagation, torch.optim.Adam(lr) (betas .9/.999, eps 1e-8).  The trainer updates
 * the weights of the predictor handle it was created on in place (that handle keeps serving
 * fpc_predictor_forward / fpc_encode with the new weights). */
typedef struct fpc_trainer fpc_trainer;
FPC_API int fpc_trainer_create(fpc_predictor* p, int max_B, int max_L, fpc_trainer** out);
FPC_API void fpc_trainer_destroy(fpc_trainer* t);
/* loss_host (host pointer, may be NULL) receives the loss of this step; passing it synchronises the stream */
FPC_API int fpc_trainer_step(fpc_trainer* t, const float* feat_dev, int B, int L, double lr, float* loss_host,
                     fpc_stream s);
/* what = 0: current parameters, 1: gradients of the last step -> host arrays in torch layouts */
FPC_API int fpc_trainer_export(fpc_trainer* t, int what, const fpc_predictor_weights* out_host);

/* ---- codebook training primitives (src/quantization/cb_func.py) ----
 * 17-dimensional vectors only (cfg['code_dims'] of the production codebooks); entries <= 4096.
 * data_dev: [nv][nd] rows, float32 (data_f64 = 0: what train_cb.py:170-178 hands to the first stage) or
 * float64 (data_f64 = 1: the residual `qr - r` later stages train on, train_cb.py:191-192);
 * codebooks float64 [entries][nd]. */

/* bytes of workspace_dev fpc_cb_update needs */
FPC_API long long fpc_cb_workspace_bytes(int nv, int entries);

/* find_nearest (cb_func.py:56-68): idx_dev[i] = first entry with the smallest float64 squared distance */
FPC_API int fpc_cb_find_nearest(const void* data_dev, int data_f64, int nv, int nd, const double* cb_dev,
                        int entries, int* idx_dev, fpc_stream s);

/* update (cb_func.py:71-100): cb_out[n] = (float64 sum, in index order, of the vectors nearest to
 * cb_in[n]) / (count[n] + 1e-20); count_dev (float64 [entries]) may be NULL */
FPC_API int fpc_cb_update(const void* data_dev, int data_f64, int nv, int nd, const double* cb_in_dev, int entries,
                  double* cb_out_dev, double* count_dev, void* workspace_dev, fpc_stream s);

/* np.mean(data, 0) as vq_train takes it (cb_func.py:34): accumulated in the data's own precision */
FPC_API int fpc_cb_mean0(const void* data_dev, int data_f64, int nv, int nd, double* out_dev, fpc_stream s);

/* ---- scalar codebooks: k-means of one feature (src/train_cb.py:219-226, commented out in the reference) ----
 * sklearn.cluster.KMeans(n_clusters = k, random_state = 0).fit(v[:, None]).cluster_centers_ as scikit-learn 1.x computes it:
 * k-means++ seeding with `trials` = 2 + int(ln k) candidates per seed, Lloyd iterations with sklearn's stopping rules (the
 * labels repeat, or the summed squared centre shift <= tol; the E-step once more in the second case), n_init runs, lowest
 * inertia wins unless it is the same clustering (csrc/kmeans1d.hip restates the operations that decide an outcome).
 *   x_dev: n float64 values on the device, CENTRED (sklearn subtracts the mean first: the caller adds it to the centres);
 *   first_ids[n_init]: each run's first seed; uniforms[n_init][k - 1][trials]: the seeding's uniform draws -- the caller draws
 *     both from numpy's RandomState(0) in sklearn's order (choice(n, p = 1 / n), then uniform(size = trials) per seed; they do
 *     not depend on the data), which makes a run pick the seeds sklearn picks; both arrays on the HOST;
 *   tol: sklearn's absolute tolerance (1e-4 * mean(var(v, axis = 0))); max_iter: 300 in sklearn;
 *   centers_host[k]; inertia_host, n_iter_host, seeds_host[n_init][k] (the seeds' indices) may be NULL.
 * Synchronous (the iteration count is data dependent: the stream is synchronised once per Lloyd iteration).
 * k <= 2048, trials <= 16, n < 2^28.  Every float64 sum has one fixed association (per block of 2 048 points: thread t adds the
 * points t, t + 256, .., a halving tree over 256 threads; block sums one after the other): results are reproducible run to run, which sklearn's OpenMP
 * reductions are not; against sklearn itself the centres agree to rounding (tests: 1e-9).  With fewer distinct values than
 * clusters the surplus centres are duplicates of existing ones (sklearn's "location of the biggest cluster" rule is followed);
 * where sklearn's own rounding noise then decides a relocation, the duplicate may sit on a different existing centre. */
FPC_API int fpc_kmeans1d(const double* x_dev, long long n, int k, int n_init, int trials, const long long* first_ids,
                 const double* uniforms, double tol, int max_iter, double* centers_host, double* inertia_host,
                 int* n_iter_host, int* seeds_host, fpc_stream s);

#ifdef __cplusplus
}
#endif
#endif /* FPCODEC_H */
