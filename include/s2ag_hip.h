/*
 * s2ag_hip.h -- C ABI of libs2ag_hip.so: the MI355X (gfx950) kernels of the Speech2AffectiveGestures
 * GAN training step.
 *
 * The reference (UttaranB127/speech2affective_gestures) is pure Python/PyTorch and has no FFI of its
 * own, so these entry points replace the ATen operators that the reference's hot path
 * dispatches.  Each declaration cites the reference call site(s) it stands in for (paths relative to
 * the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked "host".
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream, allocates
 *     nothing and never synchronises (safe under hipGraph capture).
 *   - all activations are fp32, row-major, CHANNELS-LAST: a (clips, frames, channels) tensor is a
 *     matrix of clips*frames rows; `ld*` is the row pitch in floats (lets a call read/write a column
 *     slice of a wider matrix).
 *   - weights stay in the reference's state_dict layout: conv (Cout, Cin, k), linear (out, in).
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative S2AG_E_* code.
 *   - random sites (dropout, re-parametrisation noise) are counter based: the keep mask / normal
 *     deviate of element i of site s is a pure function of (rng[0] = seed, rng[1] = step counter, s, i),
 *     where `rng` is a 2-word device array.  s2ag_dropout_mask / s2ag_normal_noise materialise the
 *     exact values a kernel will use, which is how the parity tests feed identical noise to the oracle.
 */
#ifndef S2AG_HIP_H
#define S2AG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a signature, a struct layout or the meaning of an argument changes (2, r06: s2ag_set_deterministic takes
 * two arguments, BF16Tcn / Tcn32 lost their emb_* fields, s2ag_wave12_bwd's dz_f32 == 2 mode is gone, s2ag_set_split_pieces
 * returns the override, s2ag_gen_loss takes out_rand == NULL).  _lib.py refuses a library of another version. */
#define S2AG_ABI_VERSION 2

#define S2AG_E_BADARG (-1)
#define S2AG_E_UNSUPPORTED (-2)

/* activation kinds for fused epilogues */
#define S2AG_ACT_NONE 0
#define S2AG_ACT_LEAKY 1   /* x>0 ? x : slope*x   (slope 0 = ReLU, slope 1 = identity) */
#define S2AG_ACT_SIGMOID 2

int s2ag_abi_version(void);

/* Run-time options of the library.  The library never reads the environment: the one registry of switches
 * (speech2affective_gestures_amd/config.py: name, default, what it selects, the test that arms it) pushes them here when
 * the library is loaded, and tests flip them in-process.  Nothing in the reference corresponds (it has no native code);
 * what the options select are alternative kernels for the reference call sites cited at the entry points they affect.
 *   "GRU_SPLIT"      0 f32 MFMA | 1 | 2 (default) | 3 bf16 pieces per fp32 operand (nn.GRU, net/multimodal_context_net_v2.py:480)
 * set: returns the previous value, S2AG_E_BADARG for an unknown name / negative value.  get: the value or S2AG_E_BADARG. */
int s2ag_set_option(const char* name, int value);
int s2ag_get_option(const char* name);
/* Deterministic mode: a BUILD FLAVOUR (libs2ag_hip_det.so, -DS2AG_DET=1; python -m speech2affective_gestures_amd.build --det),
 * not a run-time word in the release kernels.  s2ag_det_flavour() = 1 in that library, 0 in the release one, where
 * s2ag_set_deterministic(non-null, ..) returns S2AG_E_UNSUPPORTED and the accumulating kernels are the plain ones.
 * `zero_device_word` = one int32 device word holding 0 (null switches the mode off), `error_word` = the sticky error word
 * the caller reads back (bit 3 is raised when a workgroup's turn never came; may be null).  While set, every workgroup of an
 * accumulating launch (weight / bias gradients, embedding gradient, BatchNorm sums, derived-parameter flush: the fp32
 * atomicAdd sites behind loss.backward() at processor_v2.py:841,937) performs its atomics in the order of its linear
 * workgroup index, and in-workgroup LDS accumulation goes wavefront by wavefront -- two runs of a step on ONE stream give
 * bit-identical gradients and weights.  The caller serialises the passes of a step (Processor(deterministic=True) /
 * S2AG_DETERMINISTIC=1 does) and must not launch on side streams while the mode is on.  Both precision modes. */
int s2ag_det_flavour(void);
int s2ag_set_deterministic(int* zero_device_word, int* error_word);

/* 1-D convolution geometry, channels-last.  Input rows (n*Lin + pos), output rows (n*Lout + l),
 * pos = l*stride + tap*dil - pad (pad may be negative).  A Linear layer is ksize=1, Lin=Lout=1, N=rows. */
typedef struct {
    int N, Lin, Lout, Cin, Cout, ksize, stride, pad, dil;
    int ldx; /* row pitch of the input matrix  */
    int ldy; /* row pitch of the output matrix */
    int w_tap_major; /* 0: weight (Cout, Cin, k) as in the reference's state_dict;  1: (Cout, k, Cin) -- the layout the
                        weight-norm kernel and the ST-GCN fold emit, K-contiguous for the tile loader */
} s2ag_conv_geom;

/* fused epilogue of the forward conv: y = dropout(act(acc + bias)) */
typedef struct {
    int act;                         /* S2AG_ACT_*                                  */
    float slope;                     /* for S2AG_ACT_LEAKY                          */
    float drop_p;                    /* 0 = no dropout                              */
    const unsigned long long* rng;   /* device {seed, counter}; may be NULL if drop_p == 0 */
    unsigned site;                   /* random-site id                              */
} s2ag_epilogue;

/* y[(n,l), co] = epi( sum_{tap,ci} x[(n,pos), ci] * w[co,ci,tap] + bias[co] )      (fp32 MFMA implicit GEMM)
 * replaces: nn.Conv1d / nn.Linear forward -- net/multimodal_context_net_v2.py:18-27 (WavEncoder),
 * :39-48 (MFCCEncoder), :78 (TextEncoderTCN.decoder), :143-148 (AffEncoder conv3/4), :397-403 (pre_conv),
 * :273-276,:284-286,:483-485,:561-562 (Linear); net/tcn.py:19,25 (dilated causal conv; chomp = Lout);
 * net/utils/tgcn.py:56,181,200 (Conv2d of the ST-GCN blocks, after the host folds A / the vertex kernel
 * into a (V*Cout, V*Cin, kt) weight, see s2ag_spmv); the W_ih projections of nn.GRU (:281,:406,:480,:558). */
int s2ag_conv1d_nlc_fwd(const float* x, const float* w, const float* bias /*nullable*/, float* y,
                        const s2ag_conv_geom* g /*host*/, const s2ag_epilogue* e /*host, nullable*/, void* stream);

/* dx[(n,pos), ci] (+)= sum_{tap,co} gy[(n,l), co] * w[co,ci,tap];  g->ldx pitches dx, g->ldy pitches gy.
 * replaces: ConvolutionBackward / AddmmBackward (input grad) of the same call sites. */
int s2ag_conv1d_nlc_bwd_data(const float* gy, const float* w, float* dx, const s2ag_conv_geom* g /*host*/,
                             int accumulate, void* stream);

/* dw[co,ci,tap] (+)= sum_{n,l} gy[(n,l), co] * x[(n,pos), ci]   (split over rows, fp32 atomics)
 * replaces: ConvolutionBackward / AddmmBackward (weight grad); also dW_hh of nn.GRU with x = shifted h. */
int s2ag_conv1d_nlc_bwd_weight(const float* gy, const float* x, float* dw,
                               float* dbias /*nullable: dbias[co] (+)= sum_m gy[m, co], folded into the same launch*/,
                               const s2ag_conv_geom* g /*host*/, int accumulate, void* stream);

/* fp32 GEMM y = a w^T + bias on the bf16 matrix pipe from PRE-SPLIT operands (the GRU input projections -- W_ih x of
 * nn.GRU, net/multimodal_context_net_v2.py:281,406 -- the one GEMM shape of the step bound by the f32 matrix pipe).
 * s2ag_split_bf16x3: planes (3, rows, Kp) bf16, Kp = s2ag_split_k_padded(K): every value = piece0 + piece1 + piece2 exactly
 * to 2^-25 (each piece the bf16 rounding of what the previous ones left), zero padded along K.  s2ag_gemm_split_fwd
 * accumulates the piece products in fp32: three of them with the default two pieces (error vs fp64 ~3e-6 of the largest
 * output), six with S2AG_GRU_SPLIT=3 (as accurate as the f32-MFMA GEMM). */
int s2ag_split_k_padded(int K);
int s2ag_split_bf16x3(const float* x, int rows, int K, int ldx, void* planes, void* stream);
/* Forward of a stride-1 conv with tap-major weights on the same pipe: the weight arrives as planes (s2ag_split_bf16x3 of
 * the weight viewed as (Cout*ks, Cin)), the activations are split by the kernel's loader.  Epilogue and dropout-mask
 * indexing as s2ag_conv1d_nlc_fwd.  replaces: the dilated causal convs of net/tcn.py:19,25 (and any other stride-1
 * tap-major conv).  S2AG_E_UNSUPPORTED (nothing launched) outside stride 1 / Lin == Lout / Cin % 4 == 0. */
int s2ag_conv1d_nlc_fwd_split(const float* x, const void* w_planes, const float* bias /*nullable*/, float* y,
                              const s2ag_conv_geom* g /*host*/, const s2ag_epilogue* e /*host, nullable*/,
                              double* partials /*nullable: BatchNorm column-sum partials, sized by s2ag_conv_stats_rows*/,
                              int* stat_rows /*host; nullable together with partials*/, void* stream);

/* Weight gradients on the same pipe: dW (M, N) += gy^T x contracts over the rows, so both operands are split TRANSPOSED
 * (planes (3, cols, Rp), Rp = s2ag_split_k_padded(rows); `shift`: the operand row of frame t is frame t + shift of the same
 * clip of L frames, zero outside it -- dW_hh of nn.GRU pairs d(gh)_t with h_{t-1}; `colsum` (nullable, shift == 0): += the
 * column sums, i.e. the bias gradient) and s2ag_gemm_split_acc accumulates a w^T into y with the contraction split over
 * blocks (fp32 atomics). */
int s2ag_split_bf16x3_t(const float* x, int rows, int cols, int ldx, int shift, int L, void* planes, float* colsum,
                        void* stream);
int s2ag_gemm_split_acc(const void* a_planes /*(3, M, Kp)*/, const void* w_planes /*(3, N, Kp)*/, float* y, int M, int N,
                        int K, int ldy, void* stream);
int s2ag_gemm_split_fwd(const void* a_planes /*(3, M, Kp)*/, const void* w_planes /*(3, N, Kp)*/,
                        const float* bias /*nullable*/, float* y, int M, int N, int K, int ldy, void* stream);

/* Up to S2AG_MAX_WGRAD_JOBS ACCUMULATING weight (+ bias) gradients in one launch (a GRU layer's dW_ih and both directions'
 * dW_hh: each alone fills less than half of the chip's block slots).  Every job as s2ag_conv1d_nlc_bwd_weight with
 * accumulate = 1; S2AG_E_UNSUPPORTED (nothing launched) if a job is outside the straight-line kernel. */
#define S2AG_MAX_WGRAD_JOBS 4
typedef struct s2ag_wgrad_job {
    const float* gy;
    const float* x;
    float* dw;
    float* dbias; /* nullable */
    s2ag_conv_geom geom;
} s2ag_wgrad_job;
int s2ag_conv1d_nlc_bwd_weight_multi(const s2ag_wgrad_job* jobs /*host*/, int njobs, void* stream);

/* Both backward GEMMs of one stride-1 layer in ONE launch (they share gy): dx = (as s2ag_conv1d_nlc_bwd_data, not
 * accumulating), dw / dbias += (as s2ag_conv1d_nlc_bwd_weight with accumulate = 1).  Returns S2AG_E_UNSUPPORTED (nothing
 * launched) for geometries outside the straight-line kernels (strided, reference-layout multi-tap weights, clips shorter
 * than 32 frames): the caller then issues the two separate calls.
 * replaces: ConvolutionBackward / AddmmBackward of net/tcn.py:19,25, net/utils/tgcn.py:56,181,200 and the Linear layers. */
int s2ag_conv1d_nlc_bwd_pair(const float* gy, const float* w, const float* x, float* dx, float* dw,
                             float* dbias /*nullable*/, const s2ag_conv_geom* g /*host*/, void* stream);

/* out[c] (+)= sum_r x[r*ld + c]; if sq != NULL also sq[c] (+)= sum_r x^2.   (bias grads, BN batch statistics) */
int s2ag_colsum(const float* x, int rows, int cols, int ld, float* out, float* sq /*nullable*/, int accumulate,
                void* stream);
/* fp64 column sums of x and x^2 for the BatchNorm batch statistics: E[x^2]-E[x]^2 is only safe in double
 * (a nearly constant channel -- e.g. the mostly-zero seed-pose input -- cancels catastrophically in fp32). */
int s2ag_colstats_f64(const float* x, int rows, int cols, int ld, double* sum, double* sq, void* stream);

/* BatchNorm over a channels-last matrix whose COLUMNS map onto BN channels through chan_of_col
 * (identity for BatchNorm1d on (B,C,L); many-to-one for BatchNorm2d on the folded ST-GCN layout).
 * replaces: nn.BatchNorm1d/2d -- net/multimodal_context_net_v2.py:19,22,25 (Wav), :40-46 (MFCC),
 * :128,:139,:144,:149 (AffEncoder), :398,:401 (pre_conv); net/utils/tgcn.py:180,189,206.
 *
 * s2ag_bn_coeffs: training != 0: from fp64 column sums/sumsq (s2ag_colstats_f64) compute per-channel batch mean and
 *   biased variance, update running_mean/var (momentum, unbiased var) and num_batches_tracked (int64),
 *   and emit per-COLUMN scale/shift/mean/invstd.  training == 0: coefficients from the running statistics. */
int s2ag_bn_coeffs(const double* colsum, const double* colsq, const int* chan_of_col, int ncols, int nchan,
                   int rows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   long long* num_batches_tracked /*nullable*/, float eps, float momentum, int training,
                   float* scale_col, float* shift_col, float* mean_col, float* invstd_col, void* stream);
/* Training-mode statistics + coefficients in ONE launch (= s2ag_colstats_f64 + s2ag_bn_coeffs(training=1), without
 * the accumulator clear): row blocks write fp64 partial column sums to `partials` (2 * s2ag_bn_partial_rows(rows,
 * cols, ld) * cols doubles, contents irrelevant on entry), the block that finishes last folds them and writes the
 * coefficients.  `ticket`: one int that is 0 on entry and is left 0 on return (self re-arming); two launches that may
 * run concurrently must not share a ticket word. */
int s2ag_bn_partial_rows(int rows, int cols, int ld);
int s2ag_bn_fwd_stats(const float* x, int rows, int cols, int ldx, const int* chan_of_col /*nullable*/, int nchan,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      long long* num_batches_tracked /*nullable*/, float eps, float momentum,
                      int repeat /* >= 1: advance the running estimates (and the batch counter) this many times -- the
                                    forward passes of one step that see the same input and weights share one launch */,
                      double* partials, int* ticket, float* scale_col, float* shift_col, float* mean_col,
                      float* invstd_col, void* stream);
/* backward stages (1)+(2) below in one launch; `partials`: 2 * s2ag_bn_partial_rows * cols floats */
int s2ag_bn_bwd_stats(const float* x, const float* dy, int rows, int cols, int ldx, int lddy, const float* scale_col,
                      const float* shift_col, const float* mean_col, const float* invstd_col, float slope,
                      const int* chan_of_col /*nullable*/, int nchan, float* dgamma, float* dbeta, int accumulate,
                      float* partials, int* ticket, float* c1_col, float* c2_col, void* stream);
/* A layer that is followed by a training-mode BatchNorm can leave the batch statistics behind: the forward kernel
 * writes per-row-block column sums of its output and of its square to `partials` ((2, *stat_rows, Cout) doubles, sized
 * with s2ag_conv_stats_rows) and s2ag_bn_fold turns them into the coefficients -- no separate pass over the matrix.
 * *stat_rows == 0 on return: this geometry has no statistics epilogue, use s2ag_bn_fwd_stats. */
int s2ag_conv_stats_rows(const s2ag_conv_geom* g /*host*/);
int s2ag_conv1d_nlc_fwd_stats(const float* x, const float* w, const float* bias, float* y, const s2ag_conv_geom* g,
                              const s2ag_epilogue* e /*host, nullable*/, double* partials, int* stat_rows /*host*/,
                              void* stream);
int s2ag_bn_fold(double* partials /*consumed: large sets are pre-folded in place*/, int partial_rows, int rows, int cols, const int* chan_of_col /*nullable*/,
                 int nchan, const float* gamma, const float* beta, float* running_mean, float* running_var,
                 long long* num_batches_tracked /*nullable*/, float eps, float momentum, int repeat, float* scale_col,
                 float* shift_col, float* mean_col, float* invstd_col, void* stream);
/* OPT-IN VARIANT of s2ag_bn_fold + s2ag_bn_apply in one launch (csrc/bn_foldapply.hip; config switch BN_FOLD_APPLY): every
 * workgroup folds the (2, partial_rows, cols) sums itself, in a fixed order, workgroup 0 writes the running estimates, the
 * batch counter and the four coefficient vectors; y = leaky(x * scale + shift, slope).  Same reference call sites as
 * s2ag_bn_fold (native_batch_norm + leaky_relu_ behind a conv: net/multimodal_context_net_v2.py:18-27,39-48,397-403).
 * `partials` is only read.  supported: partial_rows * cols <= 16 384, cols <= 1 024, nchan <= 256. */
int s2ag_bn_fold_apply_supported(int partial_rows, int cols, int nchan);
int s2ag_bn_fold_apply(const double* partials, int partial_rows, int rows, int cols, const int* chan_of_col /*nullable*/,
                       int nchan, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       long long* num_batches_tracked /*nullable*/, float eps, float momentum, int repeat, float* scale_col,
                       float* shift_col, float* mean_col, float* invstd_col, const float* x, int ldx, float slope, float* y,
                       int ldy, void* stream);
/* ONE-launch BatchNorm (training): statistics, fold, coefficients AND the apply (forward: y = leaky(x*scale + shift);
 * backward: dx) in the same kernel -- the workgroups wait for the one that folds and then process the rows they have
 * just read.  replaces the same nn.BatchNorm1d/2d (+ LeakyReLU) call sites as s2ag_bn_fwd_stats + s2ag_bn_apply /
 * s2ag_bn_bwd_stats + s2ag_bn_bwd_apply (net/multimodal_context_net_v2.py:20-26,41-47,103-148, net/utils/tgcn.py:185-204)
 * for tensors of up to 4 Mi elements (s2ag_bn_fused_supported); `bar` = 3 ints, zero once, left zero; partials sized by
 * s2ag_bn_fused_partial_rows.  A wait that times out ORs 4 into the word registered with s2ag_bn_set_error_flag. */
int s2ag_bn_fused_supported(int rows, int cols);
int s2ag_bn_fused_partial_rows(int rows, int cols, int ld);
int s2ag_bn_set_error_flag(int* flag /*device, nullable*/);
int s2ag_bn_fwd_fused(const float* x, int rows, int cols, int ldx, const int* chan_of_col, int nchan, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, long long* nbt, float eps, float momentum,
                      int repeat, double* partials, int* bar, float* scale_col, float* shift_col, float* mean_col,
                      float* invstd_col, float slope, float* y, int ldy, void* stream);
int s2ag_bn_bwd_fused(const float* x, const float* dy, int rows, int cols, int ldx, int lddy, const float* scale_col,
                      const float* shift_col, const float* mean_col, const float* invstd_col, float slope,
                      const int* chan_of_col, int nchan, float* dgamma, float* dbeta, int accumulate, float* partials,
                      int* bar, float* c1_col, float* c2_col, float* dx, int lddx, void* stream);
/* y = leaky(x*scale_col + shift_col, slope) */
int s2ag_bn_apply(const float* x, int rows, int cols, int ldx, const float* scale_col, const float* shift_col,
                  float slope, float* y, int ldy, void* stream);
/* backward, 3 stages: (1) column sums of dpre and dpre*xhat, dpre = dy*leaky'(pre);
 * (2) fold columns into channels: dgamma/dbeta (+= if accumulate) and per-column c1 = S1/n, c2 = S2/n;
 * (3) dx = scale_col * (dpre - c1 - xhat*c2). */
int s2ag_bn_bwd_reduce(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                       const float* scale_col, const float* shift_col, const float* mean_col,
                       const float* invstd_col, float slope, float* s1_col, float* s2_col, void* stream);
int s2ag_bn_bwd_coeffs(const float* s1_col, const float* s2_col, const int* chan_of_col, int ncols, int nchan,
                       int rows, float* dgamma, float* dbeta, int accumulate, float* c1_col, float* c2_col,
                       void* stream);
int s2ag_bn_bwd_apply(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                      const float* scale_col, const float* shift_col, const float* mean_col,
                      const float* invstd_col, float slope, const float* c1_col, const float* c2_col, float* dx,
                      int lddx, void* stream);

/* y = leaky(a + b, slope) on column slices; b may be NULL (y = leaky(a)).
 * replaces: residual adds -- net/tcn.py:46, net/utils/tgcn.py:216-218, the bidirectional sum
 * net/multimodal_context_net_v2.py:542,:578. */
int s2ag_add_act(const float* a, int lda, const float* b /*nullable*/, int ldb, float* y, int ldy, int rows,
                 int cols, float slope, void* stream);
/* g = dy * dropmask(site) * act'(y)  -- backward of the conv epilogue / of s2ag_add_act (drop_p = 0). */
int s2ag_epilogue_bwd(const float* dy, int lddy, const float* y, int ldy, float* g, int ldg, int rows, int cols,
                      const s2ag_epilogue* e /*host*/, void* stream);

/* out[r, :] = table[ids[r], :] * dropmask;   replaces nn.Embedding + nn.Dropout --
 * net/multimodal_context_net_v2.py:70-73,:80,:88 and the speaker embedding :272,:471. */
int s2ag_embedding_fwd(const long long* ids, const float* table, int rows, int dim, int n_entries, float* out,
                       int ldo, const s2ag_epilogue* e /*host, nullable: only drop_p/rng/site used*/, void* stream);
/* dtable[ids[r], :] (+)= g[r, :] * dropmask   (dense gradient, fp32 atomics) */
int s2ag_embedding_bwd(const long long* ids, const float* g, int ldg, int rows, int dim, int n_entries,
                       float* dtable, int accumulate, const s2ag_epilogue* e /*host, nullable*/, void* stream);

/* torch.nn.utils.weight_norm (dim 0): w[r,:] = g[r] * v[r,:] / ||v[r,:]||;  net/tcn.py:19,25. */
/* ksize > 1: v is (rows, cols/ksize, ksize) and w / dw are written / read tap-major (rows, ksize, cols/ksize);
 * ksize <= 1: same layout in and out. */
int s2ag_weight_norm_fwd(const float* v, const float* g, int rows, int cols, int ksize, float* w, float* norm,
                         void* stream);
int s2ag_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, int rows, int cols,
                         int ksize, float* dv, float* dg, void* stream);

/* Several derived-parameter maps per launch (one ST-GCN block's six folds, all weight-normed convs of a TCN).
 * flush == 0: forward (y = M x;  w = g v/||v||).
 * flush != 0: route the STAGED gradient back and clear the stage: x += M^T y, y = 0   (x = gradient of the source
 *   tensor, y = staged gradient of the derived tensor);  dv += ..., dg += ..., dw = 0 for weight norm.
 * The stage is a persistent buffer that the weight-gradient kernels accumulate into (accumulate = 1): it is zero
 * before the first use and every flush leaves it zero, so no clearing launch is ever needed. */
/* Batch decode on the device.  replaces: processor_v2.py:603-606 / :614-617 (`audio * audio_max / 32767` in float64 on the
 * host, `.float()`, then the copy): the raw int16 waveform (rows, cols) and the per-clip float64 peak cross PCIe instead,
 * out = float(double(a) * peak[row] / 32767) -- bit-identical to the host path. */
int s2ag_audio_decode(const short* audio_i16, const double* peak, float* out, int rows, int cols, void* stream);
/* replaces the `.float()` of vec_seq (float64, processor_v2.py:602) and mfcc_features (float16, :607) */
int s2ag_to_f32(const void* x, int src_is_f16 /*else float64*/, float* out, long long n, void* stream);

/* diagnostics: *out = device wall clock (100 MHz ticks) when the stream reaches this point (capturable) */
int s2ag_timestamp(unsigned long long* out, void* stream);
/* diagnostics (host only, no device work): print a native back trace to file descriptor `fd` when the process dies of
 * SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL, then die as before.  The reference has no counterpart (its step is eager
 * PyTorch, processor_v2.py:776-957); the product replays multi-stream hipGraphs and this is the debug aid for faults
 * below `hipGraphLaunch` (tools/stress_starts.py; enabled with S2AG_CRASH_TRACE=1). */
int s2ag_install_crash_handler(int fd);

#define S2AG_MAX_JOBS 8
typedef struct s2ag_spmv_job {
    const int* rowptr;
    const int* col;
    const float* val;
    float* x;   /* forward: source (read);  flush: gradient of the source (+=) */
    float* y;   /* forward: derived tensor (written);  flush: staged gradient (read, then cleared) */
    int nrows;
} s2ag_spmv_job;
typedef struct s2ag_wn_job {
    const float* v;
    const float* g;
    float* w;      /* forward: out */
    float* norm;   /* forward: out;  flush: in */
    float* dw;     /* flush: staged gradient of w (read, then cleared) */
    float* dv;     /* flush: += */
    float* dg;     /* flush: += */
    int rows, cols, ksize;
} s2ag_wn_job;
int s2ag_spmv_multi(const s2ag_spmv_job* jobs /*host*/, int njobs, int flush, void* stream);
int s2ag_weight_norm_multi(const s2ag_wn_job* jobs /*host*/, int njobs, int flush, void* stream);

/* y[i] (+)= sum_{j in row i} val[j] * x[col[j]]   (CSR).  Used to fold the ST-GCN adjacency / vertex kernel
 * into dense channels-last conv weights each step and to un-fold their gradients:
 * net/utils/tgcn.py:64-71 (einsum 'nkctv,kvw->nctw'), :181 (Conv2d (kt, kv)), :200 (1x1 residual). */
int s2ag_spmv(const int* rowptr, const int* col, const float* val, const float* x, float* y, int nrows,
              int accumulate, void* stream);

/* dst (cols x rows) = src (rows x cols)^T */
int s2ag_transpose(const float* src, int rows, int cols, float* dst, void* stream);

/* One bidirectional GRU layer, recurrent part (PyTorch gate order r,z,n).  nn.GRU --
 * net/multimodal_context_net_v2.py:281,:406,:480,:558 (forward :333,:425,:541,:574).
 *  gi    (B*T, 6H): W_ih x + b_ih for [forward | reverse] direction (from s2ag_conv1d_nlc_fwd)
 *  whh   (2, 3H, H): W_hh per direction in the reference's state_dict layout;  bhh (2, 3H)
 *  whhT  (2, H, 3H): W_hh^T, needed only when s2ag_gru_seq_needs_transposed(H) != 0 (the L2-streaming kernel;
 *        the register-resident small-H kernels read `whh` directly), else NULL
 *  y     (B*T, 2H): raw hidden states [forward | reverse]
 *  ydrop (B*T, 2H): nullable; y * keep-mask(site)/(1-p) -- the next layer's input in train mode
 *  gates (2, B*T, H, 4): nullable; per unit the saved (r, z, n, W_hn h + b_hn) for the backward pass (opaque to the
 *        caller: written by the forward kernels, read by the backward kernels, one 16-byte access per unit) */
int s2ag_gru_seq_needs_transposed(int H);
int s2ag_gru_seq_fwd(const float* gi, const float* whh, const float* whhT /*nullable*/, const float* bhh, float* y,
                     float* ydrop, float* gates, int B, int T, int H,
                     const s2ag_epilogue* e /*host, nullable: dropout of ydrop*/, void* stream);
/* backward through time of one layer.
 *  dy    (B*T, lddy): grad w.r.t. the layer output the consumer saw; dir d reads columns [d*dy_dir_stride, +H)
 *        (dy_dir_stride = H for a (B*T,2H) grad, 0 when the consumer summed the two directions);
 *        if e->drop_p > 0 the keep mask of `site` is re-applied (consumer saw ydrop).
 *  whh   (2, 3H, H) reference layout; y, gates as saved by the forward.
 *  dgi   (B*T, 6H): grad of gi;  dgh (2, B*T, 3H): grad of (W_hh h + b_hh) */
int s2ag_gru_seq_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                     const float* gates, float* dgi, float* dgh, int B, int T, int H,
                     const s2ag_epilogue* e /*host, nullable*/, void* stream);

/* Cooperative variant of the two calls above for large H (s2ag_gru_coop_supported(H) != 0; H = 300 on this path):
 * W_hh stays on chip (MFMA B-operands in registers); a group of ceil(H/32) workgroups shares one (direction,
 * 16-clip slice) and exchanges h_t (forward) / d(gh)_t (backward) once per step through `workspace` with
 * write-through stores of self-validating (value, step tag) cells (agent scope).  Same arguments and results as the
 * streaming kernels; `workspace` needs s2ag_gru_coop_workspace_bytes() bytes and may be uninitialised.
 * The per-step matrix products are fp32 products realised on the bf16 matrix pipe from bf16-piece splits of the fp32
 * operands (each piece the bf16 rounding of what the previous ones left), piece products accumulated in fp32:
 *   2 pieces (default): a0b0 + a0b1 + a1b0 -- products carry 16 mantissa bits; error against an fp64 GRU 1.5e-6 (y),
 *     1.9e-6 (gradients), the f32-MFMA kernels' being 3.8e-7 / 4.6e-7 (tools/diag_gru_split.py) -- ~600x inside the 1e-3 bar;
 *   3 pieces (S2AG_GRU_SPLIT=3): + a1b1 + a0b2 + a2b0 -- 24 mantissa bits, error equal to the f32-MFMA kernels';
 *   S2AG_GRU_SPLIT=0: v_mfma_f32_16x16x4_f32.
 * s2ag_gru_coop_split_pieces(): the setting in use (shared by s2ag_gemm_split_fwd). */
int s2ag_gru_coop_supported(int H);
int s2ag_gru_coop_split_pieces(void);
/* 16-clip slices one forward workgroup alternates between (2 when B > 16 with the 3-piece products: while one slice's
 * new state travels to the peers the other slice is computed; a launch then occupies 10 * ceil(B/32) * 2 CUs) */
int s2ag_gru_coop_fwd_slices(int B);
/* an override of option GRU_SPLIT for an extent (precision contexts, tests): returns the previous OVERRIDE (-1: none), which
 * is what a caller hands back to restore -- the effective count would pin the option */
int s2ag_gru_coop_set_split_pieces(int pieces /*0, 1, 2, 3; anything else (-1): back to option GRU_SPLIT*/);
int s2ag_gru_coop_split_override(void);       /* -1: none */
long long s2ag_gru_coop_workspace_bytes(int B, int T, int H, int backward);
int s2ag_gru_coop_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop, float* gates,
                      int B, int T, int H, const s2ag_epilogue* e /*host, nullable*/, void* workspace, void* stream);
/* One layer of up to four forward passes over the SAME weights in ONE launch (the trainer's three generator passes of
 * a step -- processor_v2.py:798, :823, :909 -- run layer by layer in lockstep): pass i has its own input projections
 * gi[i], outputs y[i] / ydrop[i] / gates[i] (arrays of n device pointers on the HOST; ydrop / gates nullable as a
 * whole or per pass) and noise snapshot rng[i]; drop_p and site are the layer's.  A launch is bound by the latency of
 * its T exchanges, not by its width.  _supported = 0: issue s2ag_gru_coop_fwd per pass instead. */
int s2ag_gru_coop_fwd_multi_supported(int n, int B, int H);
long long s2ag_gru_coop_fwd_multi_workspace_bytes(int n, int B, int T, int H);
int s2ag_gru_coop_fwd_multi(int n, const float* const* gi, const float* whh, const float* bhh, float* const* y,
                            float* const* ydrop, float* const* gates, int B, int T, int H, float drop_p,
                            const unsigned long long* const* rng, unsigned site, void* workspace, void* stream);
int s2ag_gru_coop_fwd_multi_error_word_offset(int n, int B, int T, int H, long long* offset /*host*/);
int s2ag_gru_coop_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                      const float* gates, float* dgi, float* dgh, int B, int T, int H,
                      const s2ag_epilogue* e /*host, nullable*/, void* workspace, void* stream);
/* byte offset inside `workspace` of the int32 word a launch sets to 1 if it timed out waiting for a peer */
int s2ag_gru_coop_error_word_offset(int B, int T, int H, int backward, long long* offset /*host*/);
/* Production error reporting: after this call every cooperative launch of the process reports a peer time-out by
 * storing 1 to `*device_word` (sticky: the library never clears it; NULL = back to the per-workspace words).  The
 * trainer appends the word to its one read-back per step and raises -- nn.GRU of the reference
 * (net/multimodal_context_net_v2.py:480-486) cannot fail silently, so neither may its replacement. */
int s2ag_gru_coop_set_error_flag(int* device_word);

/* ---- bf16 mode of the Conv1d hot path (csrc/conv_bf16.hip) ------------------------------------------------------------
 * Replaces, with activations stored as bf16 in HBM (channels-last rows, channel count padded with ZEROS to a multiple of
 * 32 where a layer is read tap by tap), fp32 accumulation on v_mfma_f32_16x16x32_bf16 and fp32 master weights, the ATen
 * calls behind nn.Conv1d / BatchNorm1d / LeakyReLU of WavEncoder (net/multimodal_context_net_v2.py:14-33) and behind the
 * weight-normed dilated convs, ReLU, Dropout and residual adds of TemporalBlock (net/tcn.py:16-46), forward and backward.
 * The reference has no reduced-precision path (hard part 5): this is BASELINE configs[1] / [3] "bf16".
 *
 * s2ag_bf16_conv: implicit GEMM  y[n, q, co] = epi(b[co] + sum_{t<ks, c<Cp} x[n, q*pos_mul + pos_off + t*pos_tap, c] * w[co, t, c])
 *   over rows q < Lq of clips n < N; a source row outside [0, Lin) reads as zero, as do channels >= Cvalid.  Forward pass,
 *   stride-1 data gradient (tap-flipped transposed weights) and -- with phases > 1 -- the poly-phase form of a strided data
 *   gradient (phase r = blockIdx.z uses the weight set w + r*w_phase, writes y rows at offset r*y_phase and only rows q with
 *   q*phases + r < q_total).  y element (n, q, co) lives at n*y_clip + q*y_row + y_off + co; channels [Cout, CoutS) are
 *   stored as zeros (padding).  Epilogue as s2ag_conv1d_nlc_fwd (mask index row*mask_cols + co); partials / stat_rows as
 *   s2ag_conv1d_nlc_fwd_stats with s2ag_bf16_conv_stats_rows(N*Lq) rows: column sums of the ROUNDED outputs. */
typedef struct {
    const void* x;              /* bf16 */
    const void* w;              /* bf16 (phases, Cout, ks, Cp) */
    const float* bias;          /* nullable, Cout entries */
    void* y;                    /* bf16, or fp32 if out_f32 */
    int N, Lq, Lin;
    long long x_clip;           /* elements between clips of x */
    int ldx;                    /* row pitch of x, multiple of 8 */
    int pos_mul, pos_off, pos_tap;
    int ks, Cp, Cvalid;         /* Cp multiple of 32, Cvalid multiple of 8 */
    int Cout, CoutS;
    long long y_clip;
    int y_row, y_off;
    int out_f32;
    int phases;                 /* >= 1 */
    long long w_phase;
    int y_phase, q_total;
    int mask_cols;              /* 0: Cout */
    /* optional, data-gradient launches (bf16 output, phases == 1): the stored value is additionally multiplied by
     * act'(post_y) * dropout mask of the layer that produced this launch's gradient target -- post_y is that layer's
     * output, laid out exactly like y here; mask index row*post_cols + co; channels >= post_cols become zero */
    const void* post_y;
    int post_act, post_cols;
    float post_slope, post_drop;
    const void* post_rng;
    unsigned post_site;
} s2ag_bf16_conv_args;
int s2ag_bf16_conv_stats_rows(int rows);
int s2ag_bf16_conv(const s2ag_bf16_conv_args* c, const s2ag_epilogue* e /*host, nullable*/, double* partials, int* stat_rows,
                   void* stream);
/* dw[co*d_co + t*d_t + c*d_c] += sum_{n, q} gy[(n, q), co] * x[n, q*pos_mul + pos_off + t*pos_tap, c]   (co < Cout, c < Cin),
 * db[co] += sum gy[:, co] (nullable).  flat_cin > 0: the window is ONE tap of ks_out*flat_cin contiguous channels (Cp =
 * that rounded up to 64) and channel k of it is (t, c) = (k / flat_cin, k % flat_cin).  Cp multiple of 64. */
typedef struct {
    const void* gy;             /* bf16 (N*Lq, ldg) */
    const void* x;              /* bf16 */
    float* dw;
    float* db;
    int N, Lq, Lin;
    long long x_clip;
    int ldx, ldg;
    int pos_mul, pos_off, pos_tap;
    int ks, Cp, Cvalid;
    int Cout, Cin;
    long long d_co;
    int d_t, d_c;
    int flat_cin, ks_out;
} s2ag_bf16_wgrad_args;
int s2ag_bf16_conv_wgrad(const s2ag_bf16_wgrad_args* g, void* stream);
/* The same gradient as two launches without atomics: the contraction is cut into many short pieces whose tiles are stored
 * to `scratch` (s2ag_bf16_conv_wgrad_scratch_floats floats), then summed into dw / db by one thread per element -- for
 * layers with a long contraction and a small, reference-layout (scattered) weight: the wave encoder's conv2-4. */
long long s2ag_bf16_conv_wgrad_scratch_floats(const s2ag_bf16_wgrad_args* g);
int s2ag_bf16_conv_wgrad_split(const s2ag_bf16_wgrad_args* g, float* scratch, long long scratch_floats, void* stream);
/* fp32 master weights -> bf16 operand layouts, up to 32 tensors per launch (once per optimizer step):
 * dst[(o*taps + t)*Cp + c] = (c < cols and 0 <= tap0 + t*tap_step < src_taps) ? src[o*s_o + (tap0 + t*tap_step)*s_t + c*s_c] : 0 */
#define S2AG_BF16_MAX_PACK 32
typedef struct {
    const float* src;
    void* dst;
    int rows, taps, Cp, cols;
    int tap0, tap_step, src_taps;
    long long s_o;
    int s_t, s_c;
    int flat_cin;               /* > 0 (taps == 1): channel k of the single tap is (tap, c) = (k / flat_cin, k % flat_cin) */
} s2ag_bf16_pack_job;
int s2ag_bf16_pack_weights(const s2ag_bf16_pack_job* jobs /*host*/, int njobs, void* stream);
/* (rows, cols) fp32 <-> bf16 with row pitches; to_bf16 also zero-fills the pad columns [cols, ldy) */
int s2ag_bf16_cast(const void* x, int ldx, long long rows, int cols, void* y, int ldy, int to_bf16, void* stream);
/* BatchNorm apply / backward on bf16 rows (cols, ld multiples of 8); coefficient vectors are fp32 as in s2ag_bn_apply.
 * s2ag_bf16_bn_bwd: dgamma / dbeta (nullable) are accumulated atomically; `sums` is a 2*cols fp32 scratch, zero on entry
 * and exit; c1 / c2 receive the per-column means of d and d*xhat. */
int s2ag_bf16_bn_apply(const void* x, long long rows, int cols, int ld, const float* scale, const float* shift, float slope,
                       void* y, void* stream);
int s2ag_bf16_bn_bwd(const void* x, const void* dy, long long rows, int cols, int ld, const float* scale, const float* shift,
                     const float* mean, const float* invstd, float slope, float* dgamma, float* dbeta, float* sums, float* c1,
                     float* c2, void* dx, void* stream);
/* y = leaky(a + b) over n bf16 elements (b nullable; n multiple of 8);  g = dy * act'(y) * mask (pad columns zero) */
int s2ag_bf16_add_act(const void* a, const void* b, long long n, float slope, void* y, void* stream);
int s2ag_bf16_epilogue_bwd(const void* dy, const void* y, long long rows, int cols, int ld, const s2ag_epilogue* e, void* g,
                           void* stream);
/* nn.Embedding + Dropout -> bf16 rows (pad columns zero); its table gradient from bf16 dy */
int s2ag_bf16_embedding_fwd(const long long* ids, const float* table, long long rows, int dim, int n_entries, void* out, int ld,
                            const s2ag_epilogue* e, void* stream);
int s2ag_bf16_embedding_bwd(const long long* ids, const void* dy, int ld, long long rows, int dim, int n_entries, float* dtable,
                            const s2ag_epilogue* e, void* stream);
/* the one-input-channel wave conv (conv1) writing bf16 / reading a bf16 output gradient */
int s2ag_bf16_conv_c1_fwd(const float* x, const float* w, const float* bias, void* y, const s2ag_conv_geom* g, double* partials,
                          int* stat_rows, void* stream);
int s2ag_bf16_conv_c1_wgrad(const void* gy, const float* x, float* dw, float* db, const s2ag_conv_geom* g, void* stream);
int s2ag_bf16_conv_c1_rows(const s2ag_conv_geom* g);      /* statistics partial rows s2ag_bf16_conv_c1_fwd writes */

/* ---- wave encoder with BatchNorm folded into the neighbouring convs, bf16 mode (csrc/wave_fused.hip) ------------------
 * Replaces, for the training-mode WavEncoder (net/multimodal_context_net_v2.py:14-33: feat_extractor.{3,4,5,6,7,8,9} and the
 * backward of feat_extractor.{0..9}), the cudnn_convolution / native_batch_norm / leaky_relu_ and their backward ATen calls:
 * a conv stores its RAW output (bf16) + fp64 column-sum partials; the next conv applies a = leaky(scale * y + shift) in its
 * loader; backward: the data gradient of conv i+1 emits dz_i = da_i * leaky'(.) and the column sums of dz_i, dz_i * xhat_i;
 * the consumers of dy_i form dy_i = A dz_i + C y_i + B in their loaders (coefficients from s2ag_wave_bn_bwd_fold).
 * Shapes: (Cin, Cout) in {(16, 32), (32, 64), (64, 32)}, 15 taps, stride 6, no padding; x / y channels-last bf16. */
int s2ag_wave_fwd_rows(int N, int Lout, int Cin, int Cout);      /* statistics partial rows of s2ag_wave_conv_fwd */
/* the fold of a conv's statistics partials into the coefficients of the BatchNorm behind it (what s2ag_bn_fold does as a
 * launch of its own, native_batch_norm's running-estimate update included), done by the workgroup that finishes last, in
 * two levels (groups of 16 partial rows).  With R = the partial rows of the launch: `ticket` points at 1 + ceil(R / 16)
 * zero words (left zero) and the statistics buffer holds (2, R + ceil(R / 16), C) doubles. */
typedef struct s2ag_bn_fold_args {
    int* ticket;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* num_batches_tracked;   /* nullable */
    float eps, momentum;
    int repeat;                       /* running-estimate updates (>= 1), as s2ag_bn_fold */
    float* scale;                     /* out: per-channel scale, shift, mean, invstd */
    float* shift;
    float* mean;
    float* invstd;
} s2ag_bn_fold_args;
/* w_packed: (Cout, KP) bf16, k = tap * Cin + ci, zero beyond 15 Cin (bf16.WeightPack 'fwd' layout of a reference weight);
 * stats: (2, rows, Cout) doubles or NULL; fold (nullable): fold them in this launch; out_f32: y is fp32 (the last conv). */
int s2ag_wave_conv_fwd(const void* x, const float* in_scale, const float* in_shift, float slope, const void* w_packed, int KP,
                       const float* bias, void* y, int out_f32, double* stats, const s2ag_bn_fold_args* fold, int N, int Lin,
                       int Lout, int Cin, int Cout, void* stream);
/* conv1 (Conv1d(1, 16, 15, stride 5, padding 1600), net/multimodal_context_net_v2.py:18) with bf16 output, ONE statistics
 * row per block and the same in-launch fold; partials: (2, s2ag_wave_conv1_fwd_rows, 16) doubles */
int s2ag_wave_conv1_fwd_rows(const s2ag_conv_geom* g);
int s2ag_wave_conv1_fwd(const float* x, const float* w, const float* bias, void* y, const s2ag_conv_geom* g, double* partials,
                        const s2ag_bn_fold_args* fold, void* stream);

/* data gradient of conv i (poly-phase) + the backward of BatchNorm i-1 / LeakyReLU on its result: reads dy_i = ca dz + cc y
 * + cb (or the fp32 output gradient when g_f32: the last conv), writes dz_{i-1} = da_{i-1} * leaky'(p_scale y_prev + p_shift)
 * and the partial column sums (2, rows, Cin) of dz_{i-1} and dz_{i-1} * xhat_{i-1}.  w_phases: (6, Cin, 3, CPO) bf16 =
 * W[co][ci][r + 6 i] (bf16.WeightPack 'phases' layout).  ticket != NULL (a zero word that is left zero): the workgroup that
 * finishes last also does the work of s2ag_wave_bn_bwd_fold (dgamma / dbeta += ..., out_ca / out_cb / out_cc): 1 + ceil(R / 16)
 * ticket words and (2, R + ceil(R / 16), Cin) doubles of `stats` as for s2ag_bn_fold_args. */
int s2ag_wave_dgrad_rows(int N, int Lin, int Cin);
int s2ag_wave_conv_dgrad(const void* dz, const void* y, const float* ca, const float* cb, const float* cc, int g_f32,
                         const void* w_phases, int CPO, const void* y_prev, const float* p_scale, const float* p_shift,
                         const float* p_mean, const float* p_invstd, float slope, void* dz_prev, double* stats, int* ticket,
                         const float* p_gamma, float* dgamma, float* dbeta, float* out_ca, float* out_cb, float* out_cc, int N,
                         int Lin, int Lout, int Cin, int Cout, void* stream);
/* weight (+ bias) gradient of conv i: dw (Cout, Cin, 15) += sum dy_i a_{i-1}, a_{i-1} = leaky(p_scale y_prev + p_shift)
 * recomputed in the loader; partials: s2ag_wave_wgrad_blocks * Cout * 15 * Cin floats, partials_b: blocks * Cout. */
int s2ag_wave_wgrad_blocks(int N, int Lout, int Cin, int Cout);
int s2ag_wave_conv_wgrad(const void* dz, const void* y, const float* ca, const float* cb, const float* cc, int g_f32,
                         const void* y_prev, const float* p_scale, const float* p_shift, float slope, float* partials,
                         float* partials_b, float* dw, float* db, int N, int Lin, int Lout, int Cin, int Cout, void* stream);
/* fold of the partial sums of s2ag_wave_conv_dgrad: dgamma += sum dz xhat, dbeta += sum dz (nullable), and the
 * coefficients of dy = ca dz + cc y + cb for the BatchNorm over `rows` rows */
int s2ag_wave_bn_bwd_fold(const double* partials, int partial_rows, int C, long long rows, const float* gamma, const float* mean,
                          const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb, float* cc, void* stream);
/* conv1's weight gradient (net/multimodal_context_net_v2.py:18) from dz_1 and y_1 (both bf16 (N, Lout, 16)); partials:
 * s2ag_wave_conv1_wgrad_blocks * 256 floats (per-block sums, folded in order into dw / db: no atomics) */
int s2ag_wave_conv1_wgrad_blocks(const s2ag_conv_geom* g);
int s2ag_wave_conv1_wgrad(const void* dz, const void* y1, const float* ca, const float* cb, const float* cc, const float* x,
                          float* partials, float* dw, float* db, const s2ag_conv_geom* g, void* stream);

/* ---- head of the wave encoder without its (N, L1, 16) tensor in HBM (csrc/wave12.hip) ---------------------------------
 * Replaces feat_extractor[0..3] of WavEncoder (net/multimodal_context_net_v2.py:18-21: Conv1d(1,16,15,stride 5,padding)
 * BatchNorm1d(16) LeakyReLU(0.3) Conv1d(16,32,15,stride 6)) in training mode, forward and backward: conv1's output is
 * recomputed from the waveform wherever it is needed (BatchNorm statistics, conv2's forward, conv2's two gradients,
 * BatchNorm's backward, conv1's weight gradient) instead of being stored.  L1 = (Lin + 2 pad - 15) / 5 + 1,
 * L2 = (L1 - 15) / 6 + 1.  `packed`: s2ag_wave12_pack_elems() bf16 elements written by s2ag_wave12_pack from the fp32
 * weights w1 (16, 1, 15) and w2 (32, 16, 15) (once per optimizer step).
 *   s2ag_wave12_stats  column sums of z1 / z1^2 -> BatchNorm 1's running estimates and scale / shift / mean / invstd
 *                      (fold, done by the workgroup that finishes last); partials: (2, rows + ceil(rows / 16), 16) doubles
 *                      with rows = s2ag_wave12_stats_rows, fold->ticket: 1 + ceil(rows / 16) zero words (left zero).
 *                      round_bf16: statistics of the bf16-rounded z1 (bf16 mode).
 *   s2ag_wave12_fwd    z2 (N, L2, 32) = conv2(leaky(scale1 z1 + shift1)) + b2 as bf16 (out_f32 = 0; bf16 mode: z1, the
 *                      activation and z2 rounded to bf16, one bf16 product) or fp32 (out_f32 = 1: nothing rounded, operands
 *                      as two bf16 pieces, three products); partials (nullable): (2, rows (+ ceil(rows / 16)), 32) doubles,
 *                      rows = s2ag_wave12_fwd_rows: column sums of z2 / z2^2; fold (nullable): BatchNorm 2's fold in the
 *                      same launch.
 *   s2ag_wave12_bwd    from dy2 = the gradient w.r.t. z2 -- fp32 rows (dz_f32 = 1) or, in bf16 mode, ca2 dz + cc2 z2 + cb2
 *                      formed from the bf16 rows dz / z2 (wave_fused.hip) --: dw2 (32, 16, 15) +=,
 *                      dw1 (16, 1, 15) += (each nullable), dgamma1 / dbeta1 += (nullable), and ca1 / cb1 / cc1 (16 each:
 *                      dz1 = ca1 du1 + cc1 z1 + cb1, kept for inspection).  0 <= slope <= 1.  The gradients of the two biases are identically
 *                      zero (each feeds a BatchNorm) and are not formed.  Scratch, with b = s2ag_wave12_bwd_blocks(N, L1, dz_f32):
 *                      part_w2 b * 7680 floats, part_s (b + ceil(b / 16)) * 528, stats
 *                      (2, b + ceil(b / 16), 16) doubles, ticket 1 + ceil(b / 16) zero words (left zero). */
typedef struct s2ag_wave12_bwd_args {
    const float* x;                   /* (N, Lin) waveform */
    const void* packed;
    const float* b1;                  /* conv1 bias (16) */
    const float* scale1;              /* BatchNorm 1: scale, shift, mean, invstd of the forward pass; gamma */
    const float* shift1;
    const float* mean1;
    const float* invstd1;
    const float* gamma1;
    float slope;
    const void* dz;
    int dz_f32;
    const void* z2;
    const float* ca2;
    const float* cb2;
    const float* cc2;
    float* part_w2;
    float* part_s;
    double* stats;
    int* ticket;
    float* dgamma1;
    float* dbeta1;
    float* ca1;
    float* cb1;
    float* cc1;
    float* dw2;
    float* dw1;
    int N, Lin, L1, L2, pad;
} s2ag_wave12_bwd_args;
int s2ag_wave12_pack_elems(void);
int s2ag_wave12_pack(const float* w1, const float* w2, void* packed, void* stream);
int s2ag_wave12_stats_rows(int N, int L1);
int s2ag_wave12_stats(const float* x, const void* packed, const float* b1, double* partials, const s2ag_bn_fold_args* fold,
                      int round_bf16, int N, int Lin, int L1, int pad, void* stream);
/* parity tests / diagnostics: signs (N, L1, 16) bytes = 1 where scale1 z1 + shift1 > 0 -- the branch every LeakyReLU(0.3)
 * behind BatchNorm 1 takes inside s2ag_wave12_fwd / s2ag_wave12_bwd (z1 is formed by the same instructions) */
int s2ag_wave12_act_signs(const float* x, const void* packed, const float* b1, const float* scale1, const float* shift1,
                          int round_bf16, unsigned char* signs, int N, int Lin, int L1, int pad, void* stream);
int s2ag_wave12_fwd_rows(int N, int L2);
int s2ag_wave12_fwd(const float* x, const void* packed, const float* b1, const float* scale1, const float* shift1, float slope,
                    const float* b2, void* z2, int out_f32, double* partials, const s2ag_bn_fold_args* fold, int N, int Lin,
                    int L1, int L2, int pad, void* stream);
int s2ag_wave12_bwd_blocks(int N, int L1, int dz_f32);
/* tests / diagnostics: at most `cap` workgroups in s2ag_wave12_bwd (0: the default, 2 per CU); returns the previous value */
int s2ag_wave12_set_bwd_block_cap(int cap);
/* diagnostics: 120 uint64 s_memtime stamps of workgroup 0 of s2ag_wave12_bwd (per step: stash begin, loads issued, phase 2
 * begin, phase 2 end, phase 3 begin); NULL switches it off */
int s2ag_wave12_set_trace(void* buf);
int s2ag_wave12_bwd(const s2ag_wave12_bwd_args* a, void* stream);

/* ---- clip-resident TemporalConvNet in bf16 mode (csrc/tcn_fused.hip) ----------------------------------------------
 * Replaces the whole stack of TemporalBlocks of net/tcn.py:16-64 (conv1 -> chomp -> ReLU -> dropout -> conv2 -> chomp ->
 * ReLU -> dropout, + residual, ReLU; kernel size 2, dilations dil[b], in == out channels C <= 320) by ONE launch
 * forward and ONE launch for the whole chain of data gradients: a workgroup owns the rows of whole clips (clips are
 * independent along time: the dilated receptive-field window of every frame lies inside its clip), keeps the current
 * activation of its clips in LDS (bf16, 320 padded channels) through all 2*n_blocks convs and streams the weights from
 * L2 in MFMA-fragment order (s2ag_bf16_tcn_pack, once per optimizer step).  Every intermediate that the backward pass
 * needs is written to HBM once (h1 = dropout(relu(conv1)) and y = relu(h2 + x) as (clips*T, 320) bf16 rows with zero
 * pad channels, rounded exactly where the layer-by-layer bf16 kernels round; of h2 = dropout(relu(conv2)) only the sign
 * bits); the backward launch reads the sign bits, carries the gradient through LDS and leaves, per conv, the gradient w.r.t.
 * its pre-activation (gp1 / gp2, the `gy` operand of s2ag_bf16_conv_wgrad_multi) in HBM.  Dropout masks are the
 * counter-based ones of s2ag_conv1d_nlc_fwd (index row*C + channel, site[2*b + j]).
 * s2ag_bf16_tcn_clips_per_block: clips one workgroup holds (0: shape unsupported -- use the layer-by-layer kernels). */
#define S2AG_TCN_MAX_BLOCKS 4
typedef struct {
    const void* x;                              /* bf16 (clips*T, 320): input of the first block */
    void* h1[S2AG_TCN_MAX_BLOCKS];              /* bf16 (clips*T, 320) each: written forward; the weight gradients' operands */
    void* sign[S2AG_TCN_MAX_BLOCKS];            /* s2ag_bf16_tcn_sign_bytes each: "h1 > 0", "h2 > 0", "y > 0" bits, written
                                                   forward, read backward */
    void* y[S2AG_TCN_MAX_BLOCKS];
    const void* wfrag;                          /* s2ag_bf16_tcn_pack output */
    const float* bias[2 * S2AG_TCN_MAX_BLOCKS]; /* conv1, conv2 of block 0, 1, ... (nullable entries) */
    int dil[S2AG_TCN_MAX_BLOCKS];
    int n_blocks, n_clips, T, C;
    float drop_p;
    const void* rng;                            /* noise snapshot (drop_p > 0) */
    unsigned site[2 * S2AG_TCN_MAX_BLOCKS];
    /* backward only */
    const void* gy;                             /* bf16 (clips*T, 320): gradient w.r.t. the last block's output */
    void* gx;                                   /* bf16 (clips*T, 320): gradient w.r.t. x */
    void* gp1[S2AG_TCN_MAX_BLOCKS];             /* bf16 (clips*T, 320): gradient w.r.t. conv1's / conv2's pre-activation */
    void* gp2[S2AG_TCN_MAX_BLOCKS];
    /* forward only, drop_p > 0: workspace of s2ag_bf16_tcn_keep_bytes bytes (the pass's dropout keep bits, generated by a
     * launch of their own in front of the forward kernel) */
    void* keep;
} s2ag_bf16_tcn_args;
int s2ag_bf16_tcn_clips_per_block(int T, int C, int ksize);
long long s2ag_bf16_tcn_sign_bytes(int n_clips, int T);  /* bytes of one block's sign buffer */
long long s2ag_bf16_tcn_keep_bytes(int n_clips, int T, int n_blocks);
long long s2ag_bf16_tcn_pack_elems(int n_convs);          /* bf16 elements of the fragment-ordered weight set */
/* w[k]: fp32 (C, 2, C) tap-major normalised weights of conv k (k = 2*block + {0, 1}); host array of device pointers */
int s2ag_bf16_tcn_pack(const float* const* w, int n_convs, int C, void* wfrag, void* stream);
int s2ag_bf16_tcn_fwd(const s2ag_bf16_tcn_args* a, void* stream);
/* diagnostics (tools/diag_tcn_trace.py): 256 x u64 device buffer that workgroup 0 fills with s_memtime stamps at its phase
 * boundaries (forward from word 0, backward from word 128); NULL switches it off.  Not on the training path. */
int s2ag_bf16_tcn_set_trace(void* buf);
int s2ag_bf16_tcn_bwd(const s2ag_bf16_tcn_args* a, void* stream);
/* ---- clip-resident TemporalConvNet forward of the fp32 step (csrc/tcn_fused32.hip) -----------------------------------
 * Replaces the forward of the four TemporalBlocks of net/tcn.py:16-64 (the 8 dilated causal convs + ReLU + dropout +
 * residuals of TextEncoderTCN, net/multimodal_context_net_v2.py:61-91) by ONE launch: a workgroup owns one clip, fp32 rows
 * resident in LDS, products from two bf16 pieces per operand with fp32 accumulation (as s2ag_conv1d_nlc_fwd_split).
 * x, h1[b], h2[b], y[b]: fp32 (clips*T, C) rows; h1 / h2 = dropout(relu(conv)) of the block's two convs, y = relu(h2 + x_b):
 * what the layer-by-layer backward kernels need.  w[k] for s2ag_tcn32_pack: fp32 (C, 2, C) tap-major normalised weights.
 * keep: workspace of s2ag_tcn32_keep_bytes (drop_p > 0).  T <= 40, 256 < C <= 320, C % 4 == 0. */
typedef struct {
    const float* x;
    float* h1[S2AG_TCN_MAX_BLOCKS];
    float* h2[S2AG_TCN_MAX_BLOCKS];
    float* y[S2AG_TCN_MAX_BLOCKS];
    const void* wfrag;
    const float* bias[2 * S2AG_TCN_MAX_BLOCKS];
    int dil[S2AG_TCN_MAX_BLOCKS];
    int n_blocks, n_clips, T, C;
    float drop_p;
    const void* rng;
    unsigned site[2 * S2AG_TCN_MAX_BLOCKS];
    void* keep;
    /* s2ag_tcn32_bwd: the whole chain of data gradients in one launch; gp1[b] / gp2[b] (fp32 (clips*T, C)) receive the
     * gradients w.r.t. the pre-activations of block b's conv1 / conv2 -- the `gy` operands of s2ag_f32_wgrad_tr */
    const float* gy;
    float* gx;
    float* gp1[S2AG_TCN_MAX_BLOCKS];
    float* gp2[S2AG_TCN_MAX_BLOCKS];
} s2ag_tcn32_args;
int s2ag_tcn32_supported(int T, int C, int ksize);
long long s2ag_tcn32_pack_elems(int n_convs);
long long s2ag_tcn32_keep_bytes(int n_clips, int n_blocks);
int s2ag_tcn32_pack(const float* const* w, int n_convs, int C, void* wfrag, void* stream);
int s2ag_tcn32_fwd(const s2ag_tcn32_args* a, void* stream);
/* Several forward passes over the SAME weights and token ids as one batch of a->n_clips = n_passes * B clips (the trainer's
 * three generator passes of a step, processor_v2.py:798, :823, :909, differ in noise only; one workgroup per clip leaves
 * half of the chip idle at B = 128): pass k = clips [k B, (k + 1) B) draws its dropout keep bits from rngs[k] (host array
 * of device pointers; a->rng is ignored) with clip indices relative to the pass -- bit-identical to the pass run alone.
 * Only clips < save_clips (the pass with autograd) leave h1 / h2 / y of every block -- those buffers need save_clips * T
 * rows --, the others only the last block's y (a->y[n_blocks - 1]: n_clips * T rows). */
int s2ag_tcn32_fwd_passes(const s2ag_tcn32_args* a, int n_passes, const void* const* rngs /*host*/, int save_clips,
                          void* stream);
int s2ag_tcn32_bwd(const s2ag_tcn32_args* a, void* stream);
/* up to 8 s2ag_bf16_conv_wgrad jobs in one launch (the TCN's eight weight gradients fill the chip together) */
#define S2AG_BF16_MAX_WGRAD_JOBS 8
int s2ag_bf16_conv_wgrad_multi(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs, void* stream);
/* The same gradients through the LDS transpose read (csrc/wgrad_tr.hip): operand tiles go to LDS row-major as loaded and
 * ds_read_b64_tr_b16 delivers the K-major MFMA operands -- no transposing loader, no atomics (tiles of the split
 * contraction are stored to `scratch` and summed by a second launch).  Up to 8 jobs per call. */
long long s2ag_bf16_conv_wgrad_tr_scratch_floats(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs);
int s2ag_bf16_conv_wgrad_tr(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs, float* scratch, long long scratch_floats,
                            void* stream);
/* The same kernel shape for fp32 operands (gy, x: fp32 rows, pitches / Cvalid multiples of 4 floats): every value is
 * split by the loader into two bf16 pieces, a tile pair costs three MFMAs (16 mantissa bits per product, fp32
 * accumulation).  replaces: the weight gradients of nn.GRU in the fp32 step -- dW_ih = dgi^T x and dW_hh = dgh^T h_prev
 * (x = y shifted by one frame: pos_off = -1 / +1), net/multimodal_context_net_v2.py:281,406,480 -- up to 8 per call. */
/* diagnostics (tools/bench_wgrad_tr32.py): 256 x u64 device buffer for s_memtime stamps of block 0; NULL = off */
int s2ag_wgrad_tr_set_trace(void* buf);
long long s2ag_f32_wgrad_tr_scratch_floats(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs);
int s2ag_f32_wgrad_tr(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs, float* scratch, long long scratch_floats,
                      void* stream);
/* the same with an explicit workgroup count (0 = the default of 96, chosen for launches that run beside a cooperative
 * recurrence; a launch that has the chip to itself wants 256) */
long long s2ag_f32_wgrad_tr_scratch_floats_n(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs, int blocks);
int s2ag_f32_wgrad_tr_n(const s2ag_bf16_wgrad_args* jobs /*host*/, int njobs, float* scratch, long long scratch_floats,
                        int blocks, void* stream);

/* Measurement aid (tools/pmc_traffic.py): touches `bytes` of `buf` with a known access pattern so the rocprofv3 counters
 * FETCH_SIZE / WRITE_SIZE can be calibrated against a known byte count in OUR access shapes: 0 = 16 B/lane coalesced
 * reads, 1 = 8 B/lane agent-scope reads (the cooperative GRU's exchange polling), 2 = 16 B/lane writes, 3 = 8 B/lane
 * agent-scope writes.  Not on the training path. */
int s2ag_calib_traffic(void* buf, long long bytes, int pattern, void* stream);

/* ---- touched-row exchange of the embedding gradient between data-parallel replicas (csrc/rows.hip) ----------------
 * Replaces, for text_encoder.embedding.weight (nn.Embedding(n_words, 300), net/multimodal_context_net_v2.py:70-73), the
 * gradient reduction nn.DataParallel performs onto GPU 0 (processor_v2.py:167-172): only the rows a batch touches travel.
 * s2ag_rows_unique: sorted unique ids of ids[0:n_tokens) into uids[0:cap) (padded with the sentinel n_entries), their
 *   number into *count; `mark` is an n_entries-int workspace that must be zero on entry and is zero again on exit;
 *   more than `cap` distinct ids ORs 2 into *overflow_flag (nullable).
 * s2ag_rows_pack: records[s] = [int bits of uids[s] | dense[uids[s], 0:dim)] (cap x (dim + 1) floats; zeros behind a sentinel).
 * s2ag_rows_merge: `gathered` = the records of all replicas, rank-major (world x cap x (dim + 1)); for every listed row
 *   dense[row] = sum over the replicas that list it, added in rank order (identical bits on every replica). */
int s2ag_rows_unique(const long long* ids, int n_tokens, int n_entries, int cap, int* mark, int* uids, int* count,
                     int* overflow_flag, void* stream);
int s2ag_rows_pack(const float* dense, const int* uids, int cap, int dim, int n_entries, float* records, void* stream);
int s2ag_rows_merge(const float* gathered, int world, int cap, int dim, int n_entries, float* dense, void* stream);

/* z = mu + eps*exp(0.5*log_var), eps ~ N(0,1) from (rng, site);  net/embedding_net.py:10-13. */
int s2ag_reparam_fwd(const float* mu, const float* log_var, int n, const unsigned long long* rng, unsigned site,
                     float* z, void* stream);
/* dmu += dz ; dlog_var += dz * eps * 0.5 * exp(0.5*log_var) */
int s2ag_reparam_bwd(const float* dz, const float* log_var, int n, const unsigned long long* rng, unsigned site,
                     float* dmu, float* dlog_var, void* stream);

/* Discriminator loss -mean(log(d_real+1e-8) + log(1-d_fake+1e-8)) and its gradient; processor_v2.py:811.
 * Either of d_real / d_fake (with its gradient output) may be NULL: the loss is the sum of the two separable terms, so
 * the real half can be back-propagated while the generator pass that produces the fake half is still running. */
int s2ag_dis_loss(const float* d_real, const float* d_fake, int B, float* loss /*1*/, float* g_real, float* g_fake,
                  void* stream);
/* Generator losses of processor_v2.py:893-937 (+ the L1 metric of :956) fused in two launches.
 *  comps[8] = {total, huber, gen_error, div_reg, kld, l1(out,target), l1(out_tri,target), 0}
 *  weights  = {loss_regression_weight, loss_gan_weight (0 during warm-up), loss_reg_weight, loss_kld_weight}
 *  gradients of `total`: g_out (B,TP), g_dis (B), g_mu (B,16), g_logvar (B,16).  scratch: B*8 floats.
 *  out_rand == NULL selects the branch without the regulariser (processor_v2.py:933-934: z_type 'none' or loss_reg_weight
 *  0): the divergence and KLD terms are not evaluated (comps[3] = comps[4] = 0, exp(log_var) is never formed, as upstream),
 *  z / z_rand / mu / log_var / g_mu / g_logvar are not touched and may be NULL, weights[2:4] are ignored. */
int s2ag_gen_loss(const float* out, const float* target, const float* out_tri /*nullable*/, const float* dis_out,
                  const float* out_rand /*nullable*/, const float* z, const float* z_rand, const float* mu, const float* log_var,
                  int B, int TP, int ZD, const float* weights /*host[4]*/, float* scratch, float* comps,
                  float* g_out, float* g_dis, float* g_mu, float* g_logvar, void* stream);

/* Evaluation metrics of forward_pass_s2ag(calculate_metrics=True): Processor.push_samples, processor_v2.py:738-774
 * (F.l1_loss :746, convert_dir_vec_to_pose utils/ted_db_utils.py:81-102 on dir + mean_dir_vec :753-758, joint MAE over the
 * frames behind the seed poses :760-766, acceleration difference np.diff(n=2) :768-771).  out / target (B, T, 27) fp32;
 * mean_dir_vec 27 doubles; sums[3] (doubles, zeroed by the call) = {sum |out - target|, sum |joint_out - joint_tgt| over
 * frames >= n_pre, sum |acc_tgt - acc_out|}: divide by B*T*27, B*(T - n_pre)*30 and B*(T - 2)*30. */
int s2ag_pose_metrics(const float* out, const float* target, const double* mean_dir_vec, int B, int T, int n_pre,
                      double* sums, void* stream);

/* torch.optim.Adam (no weight decay / amsgrad) over a flat parameter arena; processor_v2.py:215-220.
 * `step` is a device int32 holding the number of steps already taken (bumped by s2ag_counter_inc). */
int s2ag_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                   float eps, const int* step, float grad_scale, void* stream);
/* Step guard (no reference counterpart: torch.optim.Adam.step() of processor_v2.py:815,941 is unconditional): `flag` is the
 * device word the cooperative / one-launch kernels OR their time-out bits into.  While it is non-zero s2ag_adam_step and
 * the step count of s2ag_counter_inc(counter, NULL) do nothing, so a step whose activations were wrong never reaches the
 * weights; NULL removes the guard. */
int s2ag_adam_set_guard(const int* flag);
int s2ag_counter_inc(int* counter /*nullable*/, unsigned long long* rng /*nullable: rng[1] += 1*/, void* stream);
/* noise state of one forward pass: snap[0:2] = rng[0:2], then rng[1] += 1 (the reference draws fresh torch RNG per
 * F.dropout / randn call -- net/tcn.py:22,28, net/embedding_net.py:10-13; here a pass = one counter value) */
int s2ag_rng_snapshot(unsigned long long* rng, unsigned long long* snap, void* stream);
/* The snapshots of n consecutive passes in one launch: out (n + n_extra, 2) words; row i < n = {seed, counter + i}, row
 * n + j = {seed, counter + extra_offsets[j]} (a pass further down the step's pass order, drawn early: the trainer runs the
 * generator's three passes of a step -- processor_v2.py:798, :823, :909 -- side by side); then counter += n.
 * extra_offsets: host, <= 8 entries. */
int s2ag_rng_snapshots(unsigned long long* rng, unsigned long long* out, int n, const int* extra_offsets /*host*/,
                       int n_extra, void* stream);
/* pre_seq of a step (processor_v2.py:786-789: new_zeros + two slice assignments): pre (B, T, D + 1) = the first n_pre
 * frames of target (B, T, D) with 1 in the extra column, 0 elsewhere. */
int s2ag_make_pre_seq(const float* target, float* pre, int B, int T, int D, int n_pre, void* stream);
/* The decoder's input of a generator pass -- torch.cat((pre, audio, text), dim=2) and torch.cat((in_data,
 * z_context.unsqueeze(1).repeat(1, T, 1)), dim=2), net/multimodal_context_net_v2.py:522-536 (:327-331 tri-modal) -- in one
 * launch: out (rows, sum cols) = the n <= 4 sources side by side; a source with per_clip != 0 has one row per clip of T
 * frames (the speaker code).  src / cols / ld / per_clip: host arrays.  s2ag_sum_frames is the gradient of such a source:
 * dz (B, cols) = sum over the T frames of g[:, col0 : col0 + cols] (cols <= 64). */
int s2ag_concat_cols(const float* const* src, const int* cols, const int* ld, const int* per_clip, int n, float* out,
                     long long rows, int T, void* stream);
int s2ag_sum_frames(const float* g, int ldg, int col0, int cols, int B, int T, float* dz, void* stream);

/* materialise the noise a kernel will use (parity tests / debugging only) */
int s2ag_dropout_mask(const unsigned long long* rng, unsigned site, float p, long long n, float* mask, void* stream);
int s2ag_normal_noise(const unsigned long long* rng, unsigned site, long long n, float* eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S2AG_HIP_H */
