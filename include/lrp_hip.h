/* lrp_hip.h — C ABI of the MI355X-native LRP engine (liblrp_hip.so).
 *
 * Drop-in boundary for the explanation hot path of SunJiamei/LRP-ImageCaptioning.
 * The reference is pure Python and has no FFI; each entry point below names the
 * reference call it stands in for (file:line in /root/reference; E: =
 * models/explainers.py, AB: = innvestigate/analyzer/base.py, RA:/RR: =
 * innvestigate/analyzer/relevance_based/relevance_{analyzer,rule}.py).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; every `*_dev` pointer is DEVICE memory owned by the
 *    caller (e.g. a torch tensor's data_ptr()); every `*_host` pointer is host
 *    memory read before the call returns.
 *  - the library allocates only its own workspace, freed in lrp_destroy: the
 *    caches and walk buffers in lrp_create, operand copies in lrp_set_weight[_dev],
 *    the trainer's state in lrp_train_begin; a few buffers whose need depends on
 *    which entry points a caller uses (decoder scan / gradient operand packs,
 *    per-token scale records of the fp16 fast mode, per-gate dropout products)
 *    are allocated by the FIRST call that needs them and kept.  A compute call
 *    never synchronises the device or a stream; the host side may wait on an
 *    event for the PREVIOUS call's small pinned staging copy (index arrays, the
 *    tile-order table of a reverse-walk launch) before reusing that buffer.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *  - return value: 0 = LRP_OK, negative = error; text via lrp_last_error().
 *  - one handle per GPU / stream; a handle is stateful and not re-entrant
 *    (like the reference engine, whose state lives on `self`).
 *  - tensors are row-major, images / feature maps NHWC (Keras channels_last),
 *    dense weights (in_dim, out_dim), conv kernels HWIO, LSTM gate order i,f,g,o
 *    — exactly the arrays `layer.get_weights()` yields at E:264-278 / E:1000-1019.
 */
#ifndef LRP_HIP_H
#define LRP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRP_ABI_VERSION 6

enum {
  LRP_OK = 0,
  LRP_ERR_INVALID = -1,      /* bad argument (ValueError at the Python layer)          */
  LRP_ERR_STATE = -2,        /* call order violated / weights missing                  */
  LRP_ERR_HIP = -3,          /* a HIP runtime call failed                              */
  LRP_ERR_NOMEM = -4,        /* workspace allocation failed                            */
  LRP_ERR_RANGE = -5,        /* token index out of range of the caption
                                (NotImplementedError at E:538-539 / E:1181-1182)       */
  LRP_ERR_UNSUPPORTED = -6   /* encoder/decoder kind not built (NotImplementedError)   */
};

enum { LRP_DEC_ADAPTIVE = 0,   /* ExplainImgCaptioningAdaptiveAttention  E:260-666   */
       LRP_DEC_GRIDTD = 1 };   /* ExplainImgCaptioningGridTDModel        E:995-1321  */

enum { LRP_EXPLAIN_SEQUENCE = 0,   /* _explain_lstm_single_word_sequence  E:537-666 / E:1180-1321 */
       LRP_EXPLAIN_SINGLE_STEP = 1 /* _explain_lstm_single_word           E:438-535 (adaptive)    */ };

enum { LRP_ENC_VGG = 0, LRP_ENC_RESNET = 1 };

#define LRP_MAX_CONV 32

/* Encoder = a VGG-style stack cut after the last conv's ReLU (the sub-model
 * input_1 -> block5_conv3 of E:29-30): conv3x3 'same' + ReLU layers, each
 * optionally followed by a 2x2/2 max-pool.  VGG16: 13 convs, pools after
 * 2,4,7,10.  conv_name[i] is the Keras layer name used by lrp_set_weight
 * ("<name>_W" HWIO (3,3,Cin,Cout), "<name>_b" (Cout)). */
typedef struct lrp_config {
  int32_t abi_version;          /* = LRP_ABI_VERSION                                    */
  int32_t device;               /* HIP device ordinal                                   */
  int32_t decoder;              /* LRP_DEC_*                                            */
  int32_t img_h, img_w;         /* 224, 224                                             */
  int32_t n_conv;
  int32_t conv_cin[LRP_MAX_CONV];
  int32_t conv_cout[LRP_MAX_CONV];
  int32_t conv_pool_after[LRP_MAX_CONV];
  char    conv_name[LRP_MAX_CONV][32];
  int32_t L, D, H, E, V;        /* 196, 512, 512, 512, vocab  (config.py:14-15,36-40)   */
  int32_t max_images;           /* capacity of the per-image caches                     */
  int32_t max_tokens;           /* capacity of one explain call (heat-maps)             */
  int32_t max_caption_len;      /* longest caption incl. EOS (config.py:34 -> 20+1)     */
  int32_t sos_id, eos_id;       /* tokenizer ids (1-based); model column = id-1 (E:443) */
  /* ABI v2: encoder selection.  LRP_ENC_VGG uses the conv_* table above.  LRP_ENC_RESNET builds the Keras
   * ResNet-v1 bottleneck encoder (keras_applications.resnet_common; ResNet-101 = stem 64, filters 64/128/256/512,
   * blocks 3/4/23/3) cut after the last block's ReLU (conv5_block3_out, config.py:41-45); weights are named
   * "<unit>_conv_W" HWIO, "<unit>_conv_b", "<unit>_bn_gamma|beta|mean|var" with <unit> = "conv1" or
   * "conv<s>_block<b>_<0|1|2|3>" (0 = projection shortcut).  The conv_* table is ignored. */
  int32_t encoder;              /* LRP_ENC_VGG / LRP_ENC_RESNET                         */
  int32_t resnet_stem;          /* 64                                                   */
  int32_t resnet_n_stacks;      /* 4                                                    */
  int32_t resnet_filters[8];    /* 64,128,256,512                                       */
  int32_t resnet_blocks[8];     /* 3,4,23,3                                             */
} lrp_config;

typedef struct lrp_handle lrp_handle;

/* ExplainImgCaptioningAttentionModel.__init__ (E:24-40): build the engine for a
 * model geometry; allocates all device workspace. */
int lrp_create(const lrp_config* cfg, lrp_handle** out);
int lrp_destroy(lrp_handle* h);

/* Weight extraction (E:264-278, E:1000-1019; conv kernels = the image_model's
 * layers, E:29-30).  `data_host` is float32 in the Keras layout; the library
 * splits conv weights into w+ / w- (RR:256-260), packs and uploads them.
 * Names: "<conv>_W","<conv>_b", "image_features_W/_b", "global_W/_b",
 * "embedding", "output_W/_b"; adaptive: "lstm_Wi","lstm_Wh","lstm_b","Wv","Wg",
 * "V","Wx","Wh","Ws"; grid-TD: "td_Wi","td_Wh","td_b","lang_Wi","lang_Wh",
 * "lang_b","W_va","W_ha","W_a","W_x","W_h","W_s". */
int lrp_set_weight(lrp_handle* h, const char* name, const float* data_host,
                   int32_t ndim, const int64_t* shape);
/* Same, from device memory on rank-local HBM (used after an RCCL broadcast).  ABI v3: packed by device kernels — a
 * device-to-device copy of the array plus the pack / split kernels on `stream`; no host round trip and no stream
 * synchronisation, for the conv-list (VGG) encoder, both decoders and (ABI v4) the ResNet encoder's units
 * ("<unit>_conv_W/_conv_b/_bn_gamma/_bn_beta/_bn_mean/_bn_var") alike.  The caller's buffer is copied before the call
 * returns control to the stream's next work, so it may be released (stream-ordered) right after.  Host-set and device-set
 * weights may be mixed; the last setter of a name wins. */
int lrp_set_weight_dev(lrp_handle* h, const char* name, const float* data_dev,
                       int32_t ndim, const int64_t* shape, void* stream);

/* `self._image_model.predict(img_input)` (E:375, E:1097) plus everything the
 * analyzer's forward half would recompute on every analyze() call (AB:511):
 * runs the encoder once per image and caches, per conv layer, the relevance
 * gate G_l = argmax_mask_l * a_l / SafeDivide-denominator(Z+_l) (RR:274-322,
 * layers.py:446-461) and the top feature map.  images_dev: (B,img_h,img_w,3)
 * float32, BGR mean-subtracted (preprocessors.py:43-44). */
int lrp_encode_images(lrp_handle* h, const float* images_dev, int32_t B, void* stream);

/* Decoder-only use (parity tests of the E: classes with an injected feature
 * map, mirroring a stubbed `_image_model.predict`): features_dev (B,L,D). */
int lrp_set_features(lrp_handle* h, const float* features_dev, int32_t B, void* stream);
/* Copy out the cached top feature map (B,L,D). */
int lrp_get_features(lrp_handle* h, float* features_dev, int32_t B, void* stream);

/* _forward_beam_search(X, caption) (E:370-436 / E:1092-1178) for B images at
 * once: teacher-forced replay, caches ht, ct, gt, it_act, ft_act, context,
 * attention, st, beta, c_hat, xt, caption_preds on the handle.
 * captions_host: (B, max_caption_len) tokenizer ids, row b valid for
 * lengths_host[b] entries (last valid one = EOS). */
int lrp_decoder_forward(lrp_handle* h, const int32_t* captions_host,
                        const int32_t* lengths_host, int32_t B, void* stream);

/* Read a cached decoder state array (the attributes the reference leaves on
 * `self`).  Layout (B, max_caption_len+1, dim) with row 0 = zero init, float32
 * for ht/ct/gt/it_act/ft_act/st/attention/beta, float64 for context/c_hat;
 * "caption_preds" (B, max_caption_len, V) float64; "xt" (B, max_caption_len, 2E).
 * grid-TD: h1t,c1t,g1t,i1t_act,f1t_act,h2t,c2t,g2t,i2t_act,f2t_act, context,
 * st, beta, context_hat, attention (float64 where the reference has float64). */
int lrp_read_state(lrp_handle* h, const char* name, void* out_dev, size_t out_bytes, void* stream);

/* _explain_lstm_single_word_sequence(t) (E:537-666 / E:1180-1321) for n
 * (image, t) pairs at once.  img_idx_host[i] in [0,B), t_host[i] in
 * [1, len(caption_i)] (LRP_ERR_RANGE otherwise).  Outputs (device):
 *   R_feat_dev  (n, L, D)  float32   == the (1,sqrtL,sqrtL,D) return value
 *   att_dev     (n, L)     float32   attention_t (may be NULL)
 *   r_words_dev (n, max_caption_len) float64, entry j = self.r_words[j] for
 *               j < t-1 (adaptive, normalised, first dropped, E:660-665) or
 *               j < t (grid-TD, E:1320); rest 0 (may be NULL) */
int lrp_decoder_explain(lrp_handle* h, int32_t n, const int32_t* img_idx_host,
                        const int32_t* t_host, int32_t variant, float* R_feat_dev,
                        float* att_dev, double* r_words_dev, void* stream);

/* _explain_CNN(X, R) == LRPSequentialPresetA.analyze([X, R]) (E:179-181,
 * AB:478-520) for n relevance maps at once; image i uses the caches of
 * img_idx_host[i] from the last lrp_encode_images.  R_feat_dev (n,L,D),
 * R_img_dev (n,img_h,img_w,3) float32. */
int lrp_cnn_explain(lrp_handle* h, int32_t n, const int32_t* img_idx_host,
                    const float* R_feat_dev, float* R_img_dev, void* stream);

/* Fused a7->a10 chain for a batch: decoder explain + CNN explain without the
 * R_feat round trip through the caller (R_feat_dev / att_dev / r_words_dev may
 * be NULL when not wanted). */
int lrp_explain_tokens(lrp_handle* h, int32_t n, const int32_t* img_idx_host,
                       const int32_t* t_host, int32_t variant, float* R_img_dev,
                       float* R_feat_dev, float* att_dev, double* r_words_dev, void* stream);

/* ---- caption generation: incremental decoding for the beam search of explainers.py:51-120.
 * The reference re-runs the whole captioner on every partial caption at every search step; here the hypotheses'
 * decoder state lives on the device and a search step is ONE decoder step for all of them.  The beam bookkeeping
 * (BatchNLargest, completion rule) stays with the caller.
 * lrp_decoder_gen_begin: B rows = image slots of the cached features (for a beam of k per image, cache every image's
 *   features k times with lrp_set_features); zeroes the state.  Invalidates a previous lrp_decoder_forward.
 * lrp_decoder_gen_step: step s = 0 feeds SOS; for s > 0 row r continues the hypothesis of row parent_host[r] with
 *   tokenizer id word_host[r] appended (parent and child must be rows of the same image: only the recurrent state is
 *   re-parented, a row keeps its image's features).  logits_dev (B, V) float64 = the model's un-normalised scores for position s
 *   (column k = tokenizer id k+1), i.e. row s of `caption_preds`. */
int lrp_decoder_gen_begin(lrp_handle* h, int32_t B, void* stream);
int lrp_decoder_gen_step(lrp_handle* h, int32_t B, const int32_t* parent_host, const int32_t* word_host, int32_t step,
                         double* logits_dev, void* stream);
/* ABI v3.  The ranking step of the search on the device: `_log_softmax` (explainers.py:45-48) followed by
 * `np.argpartition(preds, -beam_size)[:, -beam_size:]` (explainers.py:76-78; inference.py:205-214) for `rows` rows of
 * un-normalised scores — only k (id, log p) pairs per hypothesis cross PCIe per search step, not the (beams, V) logits.
 * logits_dev (rows, V) float64; ids_dev (rows, k) int32 = model columns (tokenizer id - 1) by descending probability,
 * ties to the lower column; logp_dev (rows, k) float64.  1 <= k <= min(V, 32). */
int lrp_op_log_softmax_topk(const double* logits_dev, int32_t rows, int32_t V, int32_t k, int32_t* ids_dev,
                            double* logp_dev, void* stream);

/* ---- gradient baselines on the same caches -------------------------------------------------
 * lrp_decoder_gradient == ExplainImgCaptioning{AdaptiveAttention,GridTD}Gradient._lstm_decoder_backward(t)
 * (models/explainers.py:780-832, :1452-1532) for n (image, t) units at once: d_feat_dev (n, L, D) float32 =
 * the reference's hand-written BPTT of logit[caption[t-1]-1] w.r.t. the CNN features (its simplifications
 * included); r_words_dev (n, max_caption_len) float64 or NULL, columns >= t untouched zeros.
 * Errors as lrp_decoder_explain. */
int lrp_decoder_gradient(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const int32_t* t_host,
                         float* d_feat_dev, double* r_words_dev, void* stream);
/* lrp_cnn_walk == <Analyzer>(image_model, neuron_selection_mode="replace").analyze([X, head]) for
 * Gradient / InputTimesGradient / GuidedBackprop (innvestigate/analyzer/gradient_based.py:101-265;
 * explainers.py:672, :884, :928) — and LRPSequentialPresetA for LRP_WALK_LRP (same as lrp_cnn_explain).
 * head_dev (n, L, D) -> out_dev (n, img_h, img_w, 3).  VGG-style encoders only for the gradient walks. */
enum { LRP_WALK_LRP = 0, LRP_WALK_GRADIENT = 1, LRP_WALK_INPUT_X_GRADIENT = 2, LRP_WALK_GUIDED_BACKPROP = 3 };
int lrp_cnn_walk(lrp_handle* h, int32_t n, const int32_t* img_idx_host, const float* head_dev, float* out_dev,
                 int32_t walk, void* stream);

/* Arithmetic of the per-token reverse walk through the encoder (lrp_cnn_explain / lrp_explain_tokens).
 * LRP_PREC_BF16X3 DEFAULT.  Split-bf16 walk: every fp32 operand is carried as hi + lo bf16 (16 mantissa bits) and every
 *                 product is hi*hi' + hi*lo' + lo*hi' on v_mfma_f32_32x32x16_bf16 with fp32 accumulation — three MFMAs
 *                 in every layer, no scaling state.  Worst case per product 2^-16 (8e-6 << the 1e-4 relative-L1 bar)
 *                 whatever the weights look like; measured 3e-6 ... 7e-6 on dense Gaussian and on sparse heavy-tailed
 *                 (trained-like) kernels alike (tests/test_gpu_stress_parity.py).  The per-image forward's activations
 *                 are fp32-grade (fp16 pairs on both operands, three MFMAs, blocked accumulation: 7e-7 on the features
 *                 like the fp32 MFMA); the denominators Z+ are three-term as well.
 * LRP_PREC_FP32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere: the reference's own arithmetic (TF float32).
 * LRP_PREC_F16X2  OPT-IN fast mode (VGG-style encoders; ABI v3).  The relevance tensors are fp16 pairs hi + lo carried
 *                 scaled by a per-token power of two; layers after the last pool take the full three-MFMA product, every
 *                 layer below with >= 64 output channels reads only the hi half of the weights: ONE fp16 (11 bits) per
 *                 weight, TWO MFMAs per product, and the denominators Z+ of those layers are computed with the same
 *                 rounded weights.  ~12 % faster than the default — but the weight rounding is the same for every token
 *                 and only averages out when a sum has many comparable products: 3-5e-6 on dense Gaussian kernels,
 *                 1e-4 ... 5e-4 (OUTSIDE the bar) on sparse heavy-tailed ones.  Worst case 2^-12 = 2.4e-4 per product.
 *                 Use it only after checking it on the model at hand (bench.py reports it as `fast_mode` with its own
 *                 in-run parity).  Conv widths % 8 != 0 silently use the fp32 path; the ResNet walk runs as BF16X3.
 * LRP_PREC_BF16X3_FAST  LRP_PREC_BF16X3 with two-way split activation convs in the forward as well: faster encode, but
 *                 arg-max flips put the heat-map parity at 2e-5 ... 9e-5 (five seeds).  Opt-in, not recommended.
 * The decoder is fp32 / fp64 in every mode.  A mode CHANGE drops the encode caches (the gates were computed in the old
 * arithmetic): lrp_encode_images must run again before the next explain call (LRP_ERR_STATE otherwise). */
enum { LRP_PREC_FP32 = 0, LRP_PREC_BF16X3 = 1, LRP_PREC_BF16X3_FAST = 2, LRP_PREC_F16X2 = 3 };
int lrp_set_precision(lrp_handle* h, int32_t mode);
/* ABI v5.  Which conv layers take the two-MFMA form in LRP_PREC_F16X2 mode: bit li of `mask` = the reverse launch through
 * conv li (and the forward denominators Z+ of that layer, which always follow it) reads ONE fp16 per weight; every other
 * layer takes the three-MFMA product on fp16 pairs (22 mantissa bits on both operands — at least as exact as
 * LRP_PREC_BF16X3).  mask = -1 restores the built-in rule (all layers up to the last pool with >= 576 products per sum);
 * mask = 0 is "fp16 pairs, three MFMAs everywhere".  This is the hook for a per-MODEL decision: calibration.py measures,
 * on the caller's own weights and images, the heat-map error each layer's two-term form causes against the exact-fp32
 * mode and enables it only where the measured error leaves the requested margin below the 1e-4 bar (the rule's error
 * depends on the weight statistics: see LRP_PREC_F16X2 above).  Layer 0 (the image layer) has no two-term form
 * (LRP_ERR_INVALID), nor have bits beyond the configured convs.  A change drops the encode caches when the handle is in
 * LRP_PREC_F16X2 mode (LRP_ERR_STATE from explain calls until lrp_encode_images ran again). */
int lrp_set_fast_layers(lrp_handle* h, int64_t mask);

/* Dominant-kernel timing for bench.py's roofline block: when enabled, HIP
 * events bracket every conv-LRP launch on the caller's stream; query returns
 * launches and summed milliseconds since the last reset (syncs the events). */
int lrp_profile_enable(lrp_handle* h, int32_t on);
int lrp_profile_query(lrp_handle* h, int64_t* n_launches, double* total_ms, double* total_flop);
/* Same, per launch: fills up to `cap` entries of ms_out / flop_out (in launch order) and
 * returns the count in *n_out; the records are consumed. */
int lrp_profile_records(lrp_handle* h, int32_t cap, double* ms_out, double* flop_out, int32_t* n_out);

/* Workspace bytes held by the handle. */
int64_t lrp_workspace_bytes(const lrp_handle* h);

/* Operator-level entry (unit tests of the dominant kernel): implicit-GEMM
 * 3x3 'same' (taps=9) or 1x1 (taps=1) convolution on NHWC float32 with the
 * fp32 MFMA path.  w_hwio_host: (3,3,Cin,Cout) or (1,1,Cin,Cout).
 * mode 0: out = relu(conv + bias); mode 1: out = conv + bias;
 * mode 2: out = convT(in, w) * aux   (LRP backward with gate `aux`, shape of out)
 * mode 3: like 2 with 2x nearest up-sampling of the conv result (pool routing).
 * mode | LRP_CONV_SPLIT_BF16 (modes 1..3, channels % 8 == 0): the same operator on the split-bf16
 * MFMA path the engine uses by default (three bf16 MFMAs per fp32 product); in/out stay float32. */
enum { LRP_CONV_SPLIT_BF16 = 0x100 };
int lrp_op_conv(const float* in_dev, const float* w_hwio_host, const float* bias_host,
                const float* aux_dev, float* out_dev, int32_t NB, int32_t H, int32_t W,
                int32_t Cin, int32_t Cout, int32_t taps, int32_t mode, void* stream);

/* ABI v6.  The conv-LRP launch BEHIND a 2x2 max-pool (AlphaBetaRule RR:274-322 for a conv whose output feeds MaxPooling2D: the
 * relevance arrives through the pool's gradient routing, RA:470-480 -> IL:138-157, i.e. one non-zero per window and channel)
 * on the 2:4-sparse matrix cores (csrc/conv_sparse.h): out[n][y][x][ci] = gate[n][y][x][ci] x sum over taps, co of
 * S[n][y+dy][x+dx][co] w+[..], with S given in COMPACT form — sc_dev (NB, Hp, Wp, Cout) fp32 = the value of each window's
 * non-zero, pos_dev (same shape, bytes) = its position 2 dy + dx.  w_hwio_host (3, 3, Cin, Cout) like lrp_op_conv's backward
 * modes; gate_dev / out_dev (NB, 2 Hp, 2 Wp, Cin) fp32.  Split-bf16 arithmetic (three matrix instructions per product).
 * Needs Cin % 256 == 0, Cout % 16 == 0.  reps >= 1 repeats the launch (profiling).  Same values as lrp_op_conv mode 2 with
 * LRP_CONV_SPLIT_BF16 on the expanded tensor up to the summation order (tests/test_gpu_conv_sparse.py). */
int lrp_op_conv_pool_sparse(const float* sc_dev, const unsigned char* pos_dev, const float* w_hwio_host, const float* gate_dev,
                            float* out_dev, int32_t NB, int32_t Hp, int32_t Wp, int32_t Cin, int32_t Cout, int32_t reps, void* stream);

/* Operator-level entries for the other rules LRPSequentialPresetA can reach (not on the
 * truncated-VGG16 path; ResNet-101 "next" row).  All pointers device memory unless `_host`.
 * EpsilonRule with bias=False (RR:113-144, RA:706-711): x (N,Din), W_host (Din,Dout) Keras
 * layout, R (N,Dout) -> out (N,Din).  Din, Dout multiples of 4. */
int lrp_op_epsilon_dense(const float* x_dev, const float* W_host, const float* R_dev, float* out_dev,
                         int32_t N, int32_t Din, int32_t Dout, float epsilon, void* stream);
/* BatchNormalizationReverseLayer (RA:197-257): channels-last x, R, out of n elements, C channels;
 * gamma/beta/mean/var device vectors of length C. */
int lrp_op_batchnorm_lrp(const float* x_dev, const float* gamma_dev, const float* beta_dev, const float* mean_dev,
                         const float* var_dev, float bn_eps, const float* R_dev, float* out_dev, int64_t n,
                         int32_t C, void* stream);
/* AddReverseLayer (RA:260-286): R_a = a*SafeDivide(R,a+b), R_b = b*SafeDivide(R,a+b). */
int lrp_op_add_lrp(const float* a_dev, const float* b_dev, const float* R_dev, float* Ra_dev, float* Rb_dev,
                   int64_t n, void* stream);

/* AveragePoolingReverseLayer (RA:289-316) for k x k / stride k 'valid' average pooling, channels-last:
 * x (NB,H,W,C), R (NB,H/k,W/k,C) -> out (NB,H,W,C) = x * SafeDivide(R, avgpool(x))[window] / k^2.
 * (Not reachable from either encoder cut — VGG16 block5_conv3, ResNet conv5_block3_out — provided as an operator.) */
int lrp_op_avgpool_lrp(const float* x_dev, const float* R_dev, float* out_dev, int32_t NB, int32_t H, int32_t W,
                       int32_t C, int32_t k, void* stream);

/* Heat-map rendering of explain_image.py:55-60 (`heatmap(postprocess(relevance))`, innvestigate/examples/
 * utils_imagenet.py:31-33 -> utils/visualizations.py:87-125, :57-80): per heat-map sign-preserving gamma correction
 * (0.95 in the reference), channel sum, abs-max projection to 0..255 and a 256 x 3 colormap lookup (lut_dev, the
 * reference uses matplotlib's 'seismic').  R_img_dev (n, npix, C) -> rgb_dev (n, npix, 3) float32.  An all-zero map
 * (0/0 and an undefined NaN->int cast in the reference) renders as the mid-scale colour. */
int lrp_heatmap_render(const float* R_img_dev, const float* lut_dev, float* rgb_dev, int32_t n, int32_t npix, int32_t C,
                       float gamma, void* stream);

/* Image preprocessing (models/preprocessors.py:38-53, vgg16 / vgg19 / resnet101 = keras preprocess_input 'caffe'):
 * rgb_dev (NB, H0, W0, 3) decoded uint8 RGB -> out_dev (NB, H, W, 3) float32: nearest-neighbour resize as PIL does
 * for load_img(target_size=(H, W)), RGB -> BGR, minus the ImageNet means.  The result feeds lrp_encode_images. */
int lrp_preprocess_images(const uint8_t* rgb_dev, float* out_dev, int32_t NB, int32_t H0, int32_t W0, int32_t H,
                          int32_t W, void* stream);

/* Score reduction of the LRP-inference layer (models/model.py:1675-1686) for n heat-maps:
 * hp = mean over channels, hp /= max|hp|, then mode 0 = mean, 1 = mean(max(hp,0)), 2 = np.quantile(hp, 0.9).
 * R_img_dev (n, npix, C) float32, scores_dev (n) float64. */
int lrp_heatmap_scores(const float* R_img_dev, double* scores_dev, int32_t n, int32_t npix, int32_t C, int32_t mode,
                       void* stream);

/* Operator-level entries of the fine-tune step's dense products (unit tests at real layer sizes; csrc/train_gemm.h).
 * lrp_op_sgemm: C (+)= op(A) op(B) on the fp32 matrix cores, row-major with leading dimensions;
 *   op(A)[m][k] = transA ? A[k*lda + m] : A[m*lda + k], op(B)[k][n] = transB ? B[n*ldb + k] : B[k*ldb + n]
 *   (not both transposed).  ws_dev / ws_floats: scratch for the split over K (NULL: no split).
 * lrp_op_conv_wgrad: weight and bias gradient of a 3x3 'same' convolution: x (NB,H,W,Cin), dz (NB,H,W,Cout) ->
 *   dw (3,3,Cin,Cout) HWIO, db (Cout) or NULL; all nine taps in one launch, K = NB*H*W split deterministically. */
int lrp_op_sgemm(const float* A_dev, const float* B_dev, float* C_dev, int32_t M, int32_t N, int64_t K, int64_t lda, int64_t ldb,
                 int64_t ldc, int32_t transA, int32_t transB, int32_t accumulate, float* ws_dev, int64_t ws_floats, void* stream);
int lrp_op_conv_wgrad(const float* x_dev, const float* dz_dev, float* dw_hwio_dev, float* db_dev, int32_t NB, int32_t H, int32_t W,
                      int32_t Cin, int32_t Cout, float* ws_dev, int64_t ws_floats, void* stream);
/* the same product with bf16 operands (LRP_TRAIN_BF16; csrc/train_gemm_bf16.h) */
int lrp_op_conv_wgrad_bf16(const float* x_dev, const float* dz_dev, float* dw_hwio_dev, float* db_dev, int32_t NB, int32_t H,
                           int32_t W, int32_t Cin, int32_t Cout, float* ws_dev, int64_t ws_floats, void* stream);

/* ---- Fine-tune step of the LRP-inference training loop (train.py:573-581:
 * `keras_model.train_on_batch(X + [lrp_weight], [y, y])` on ImgCaptioningAdaptiveAttentionLRPInferenceModel,
 * models/model.py:1340-1374, and its grid-TD twin, train.py:645-656 on ImgCaptioningGridTDLRPInferenceModel,
 * models/model.py:1254-1311; VGG encoder, the decoder the handle was created with).  Per batch the caller runs lrp_encode_images,
 * computes lrp_weight with the explain entry points above, then:
 *   lrp_train_step : training-mode decoder forward on the cached features, loss 0.5 CE(y, logits[:, :-1]) +
 *                    0.5 CE(y, (logits * lrp_weight)[:, :-1]) (M:95-103, :1370-1373), backward through the decoder and
 *                    the encoder (all layers trainable, M:1332-1333) -> grads_dev, losses_dev (5 floats) = what train_on_batch
 *                    returns: total, loss head 1, loss head 2, accuracy head 1, accuracy head 2 (M:105-124)
 *   (the host all-reduces grads_dev across ranks)
 *   lrp_train_apply: Adam(lr, clipvalue) as keras does (clip element-wise, then the moments; epsilon 1e-7) on the fp32
 *                    master weights; the engine's operand copies are rebuilt, cached images are dropped.
 * All parameters live in one flat fp32 buffer: lrp_train_num_params / lrp_train_param_info give name, offset and size
 * of each slice ("<layer>_W" HWIO, "<layer>_b", then the decoder weights under their lrp_set_weight names);
 * grads_dev has the same layout, lrp_train_flat_size floats.  cap_in_dev (B, T) int32 embedding rows (the model's
 * `captions_input`), y_idx_dev (B, T) int32 class index of the one-hot label row or -1 for an all-zero row,
 * lrp_weight_dev (B, T, V) float32.  Dropout masks (values 0 or 1/(1-p)) or NULL: image_features (B, L, H),
 * global (B, E), output (B, T, H); LSTM-cell dropout (keras `dropout` / `recurrent_dropout`, one mask per gate i f c o
 * and per step because the wrapper calls the cell inside the K.rnn loop, M:582): lstm_in (T, 4, B, 2E),
 * lstm_rec (T, 4, B, H) (for grid-TD the language LSTM: lstm_in (T, 4, B, 2H)); logits (B, T, V): the grid-TD model's
 * Dropout on the logits (M:1303-1304), NULL for the adaptive model. */
int lrp_train_begin(lrp_handle* h, float lr, float clipvalue, float beta1, float beta2, float eps);
/* ABI v3.  Arithmetic of the step's convolution gradients (BASELINE config 5 names bf16; models/model.py:1340-1374 is
 * float32 in the reference).  Master weights, Adam and every accumulation are fp32 in both modes.
 *   LRP_TRAIN_FP32 (default)  weight gradients on the fp32 MFMA; backward-data convs split-bf16 (three bf16 MFMAs per
 *                             product, fp32-grade) — exact fp32 when the handle is in LRP_PREC_FP32.
 *   LRP_TRAIN_BF16            the encoder's weight gradients with bf16 operands (activations and dZ rounded to bf16,
 *                             v_mfma_f32_32x32x16_bf16, fp32 accumulate): conv weight gradients within ~1e-2 relative L1
 *                             of the fp32 ones (tests/test_gpu_train.py), 5-10x less matrix time. */
enum { LRP_TRAIN_FP32 = 0, LRP_TRAIN_BF16 = 1 };
int lrp_train_set_precision(lrp_handle* h, int32_t mode);
int64_t lrp_train_flat_size(const lrp_handle* h);
int32_t lrp_train_num_params(const lrp_handle* h);
int lrp_train_param_info(const lrp_handle* h, int32_t i, const char** name, int64_t* offset, int64_t* size);
int lrp_train_step(lrp_handle* h, int32_t B, int32_t T, const int32_t* cap_in_dev, const int32_t* y_idx_dev,
                   const float* lrp_weight_dev, const float* mask_image_features_dev, const float* mask_global_dev,
                   const float* mask_output_dev, const float* mask_lstm_in_dev, const float* mask_lstm_rec_dev,
                   const float* mask_logits_dev, float* grads_dev, float* losses_dev, void* stream);
/* Optional: the training-mode decoder forward of lrp_train_step ahead of time (it needs neither lrp_weight nor the labels),
 * e.g. on a second stream under the explanation that produces lrp_weight.  A following lrp_train_step with the same B, T
 * waits for it and starts at the loss.  Contract: the kernels read cap_in and the masks straight from the caller's device
 * buffers (nothing is staged), in the forward AND in the backward scan of lrp_train_step — so that step must be handed the
 * SAME cap_in / mask pointers (they must stay alive and unchanged until the step's work has completed); other pointers are
 * refused with LRP_ERR_INVALID rather than back-propagated through masks the forward did not use, and the refusal drops
 * the pending forward (a retry runs its own, so recycled addresses cannot pass for the old buffers).  lrp_encode_images,
 * lrp_set_features, lrp_set_weight[_dev] and lrp_set_precision drop a pending early forward. */
int lrp_train_forward(lrp_handle* h, int32_t B, int32_t T, const int32_t* cap_in_dev, const float* mask_image_features_dev,
                      const float* mask_global_dev, const float* mask_output_dev, const float* mask_lstm_in_dev,
                      const float* mask_lstm_rec_dev, void* stream);
/* ABI v6.  Forget a pending lrp_train_forward (train.py:571-577 has no counterpart: the reference runs the forward inside
 * train_on_batch).  For a caller that abandons the step it issued the early forward for — e.g. an error between the two
 * calls — and is about to release cap_in / the masks: `stream` (or, NULL, the host) first waits for the early forward, which
 * may still be reading them; the next lrp_train_step then runs its own forward.  lrp_train_step does the same by itself when
 * it refuses mismatching pointers.  No-op without a pending forward. */
int lrp_train_drop_forward(lrp_handle* h, void* stream);
int lrp_train_apply(lrp_handle* h, const float* grads_dev, void* stream);
int lrp_train_get_master(lrp_handle* h, float* flat_dev, void* stream);

/* ABI v4.  Kernel launches issued by this library since it was loaded (all handles, this process; copies and memsets not
 * counted): bench.py brackets one single-image explanation with it (`latency.launches`). */
int64_t lrp_launch_count(void);

/* ABI v6.  The library's A/B switches (LRP_CONV_HALO, LRP_UP2_PW, LRP_IMG_FOLD, ... — measurement knobs and
 * tested fall-backs, every one exercised by tests/test_gpu_switches.py) are read from the environment once per process, not on
 * the launch path; this re-reads them.  For tests and measurement scripts: call it between launches, never while another
 * thread is inside the library.  A switch that changes how lrp_encode_images lays out its caches takes effect at the next
 * lrp_encode_images.  The reference has no counterpart. */
int lrp_reload_switches(void);

const char* lrp_last_error(void);
int lrp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LRP_HIP_H */
