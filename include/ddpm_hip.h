/* ddpm_hip.h — C ABI of libddpm_hip.so: the MI355X (gfx950) kernels behind the ddpm-torch hot path.
 *
 * The reference (tqch/ddpm-torch) is pure Python on ATen; it has no FFI of its own.  Each entry point below
 * replaces the ATen op class the reference invokes at the cited site (paths relative to the upstream repo);
 * the binding a maintainer would add is the ctypes table in ddpm-torch_amd/ddpm_torch/_hip.py.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator); the library never
 *     allocates, frees or retains memory.  "stream" is a hipStream_t; all work is enqueued on it and the call
 *     returns immediately (no synchronisation, hipGraph-capturable).  Re-entrant and thread-safe.
 *   - dtype: 0 = fp32, 1 = bf16 (activations / packed weights); parameters, statistics and gradients are fp32.
 *   - Activations are NHWC: element (b, y, x, c) at  base + ((b*H + y)*W + x)*ld + c  with a caller-chosen pixel
 *     pitch ld >= C (in elements), so channel slices of a wider buffer are valid operands (zero-copy concat).
 *     Pointers and pitches must be 16-byte aligned; channel counts are multiples of 16/sizeof(dtype).
 *   - Return value: 0 OK, 1 bad shape, 2 bad dtype, 3 misaligned, 4 launch failure, 5 null pointer.
 */
#ifndef DDPM_HIP_H
#define DDPM_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* F.conv2d 3x3/1x1 (ddpm_torch/modules.py:120-123; sites models/unet.py:36,37,41,76,79,80,127,141,201), with
 * SamePad2d(3,2)+stride 2 (modules.py:153-160, unet.py:165-167), nn.Upsample(2,"nearest") (unet.py:199),
 * the per-sample time-bias add (unet.py:86) and the residual add (unet.py:60,89) fused:
 *   y[b,oy,ox,n] = sum_{r,s,c} x[b, f(oy*stride+r-pad_t), f(ox*stride+s-pad_l), c] * w[n][r][s][c]
 *                  + bias[n] + rowbias[b*rowbias_ld + n] + residual[b,oy,ox,n]      (+ y when accumulate)
 * f = identity | v>>1 (upsample=1, H/W are the stored low-res dims) | v/2 for even v only (dilate=1: the dgrad
 * of a stride-2 conv).  w is packed [N][R][S][C] in dtype (ddpm_pack_weight).  dgrad = same call on dy with the
 * flipped/transposed pack and pad = R-1-pad.  out_mode: 0 NHWC dtype (pitch y_ld) | 1 NHWC fp32 | 3 NCHW fp32.
 * splits > 1 splits K over blocks and reduces in-launch (for layers with few output tiles): splitk_ws holds
 * max(ceil(M/128)*ceil(N/128)*16384, ceil(M/64)*ceil(N/64)*4096)*splits floats and splitk_cnt one zero-initialised counter
 * per 64x64 output tile (left zero) — the library picks the tile size.  splits == 2 is an OFFER to the small-grid kernel (at most
 * 128 tiles of 64x64): it takes 2, 4 or 8 K runs per tile with tiles x runs <= 256, so for splits == 2 splitk_ws must hold at
 * least 256*4096 floats (4 MiB) as well. */
int ddpm_conv2d_nhwc(const void* x, long long x_ld, const void* w, void* y, long long y_ld,
                     const float* bias, const float* rowbias, long long rowbias_ld,
                     const void* residual, long long res_ld,
                     int B, int H, int W, int C, int Ho, int Wo, int N, int R, int S,
                     int stride, int pad_t, int pad_l, int upsample, int dilate,
                     int accumulate, int out_mode, int splits, float* splitk_ws, unsigned* splitk_cnt,
                     int dtype, void* stream);

/* convolution_backward w.r.t. the weight (autograd of the sites above), into a PACKED gradient [Nreal][R][S][Creal]:
 *   dw[n][r][s][c] += sum_{b,oy,ox} dy[b,oy,ox,n] * x[b, f(..), f(..), c]      n < Nreal, c < Creal
 * dy has N (>= Nreal) channels, x has C (>= Creal) channels (zero padding beyond the real counts).  The reduction over
 * B*Ho*Wo is cut into `splits` slices that run as separate workgroups:
 *   slab_stride == 0: every slice adds into dw with fp32 atomics (dw must be zero-filled / hold earlier contributions);
 *   slab_stride  > 0: slice s STORES its partial at dw + s*slab_stride (floats) — `splits` must then equal
 *                     ddpm_wgrad_effective_splits(B*Ho*Wo, splits, dtype) so that every copy is written — and
 *                     ddpm_wgrad_reduce sums the copies in a fixed order (bit-deterministic gradients).
 *                     table[i] = {src address of copy 0, dst address, floats, copies, stride}.
 * ddpm_wgrad_unpack rewrites all layers' packed gradients into the parameter layout [n][c][r][s] in one launch:
 * descs[i] = {src offset in gpack, dst offset in gflat (floats), N, C, R*S} as int64 (R*S = 1: plain segment copy);
 * every value is multiplied by `scale` (1/world_size after the data-parallel sum all-reduce of gpack). */
int ddpm_conv2d_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                           int B, int H, int W, int C, int Creal, int Ho, int Wo, int N, int Nreal, int R, int S,
                           int stride, int pad_t, int pad_l, int upsample, int splits, int dtype, void* stream);

int ddpm_wgrad_effective_splits(int K, int splits, int dtype);

/* The same gradient for the 3x3 / stride 1 / pad 1 sites (ResidualBlock.conv1 / conv2, ddpm_torch/models/unet.py:76,79), bf16, by a
 * patch-stationary kernel: a block keeps a 64 x 32 x 9-tap output tile and walks 16x16 pixel patches whose dy tile and x halo are
 * staged in LDS once for all nine taps; the reduction over pixels is cut into few slices.  Also produces the bias gradient
 * dbias[n] = sum_{b,y,x} dy[b,y,x,n] (optional) from the dy fragments it reads anyway.
 *   slab_stride == 0: fp32 atomics into dw [Nreal][3][3][C] / dbias [Nreal];
 *   slab_stride  > 0: slice s stores at dw + s*slab_stride (and dbias + s*bias_stride); all ddpm_conv3x3_wgrad_splits(...)
 *                     copies are written in full and ddpm_wgrad_reduce sums them in a fixed order (deterministic).
 * C % 32 == 0, N % 8 == 0 and H = W in {4, 8} or H, W multiples of 16; otherwise DDPM_ERR_SHAPE (use ddpm_conv2d_wgrad_nhwc).
 * splits <= 0: chosen by the library. */
int ddpm_conv3x3_wgrad_splits(int B, int H, int W, int C, int N, int splits);
int ddpm_conv3x3_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                            float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                            int dtype, void* stream);
/* ... of the conv inside an Upsample block (`nn.Upsample(scale_factor=2, mode="nearest")` then the 3x3 conv, ddpm_torch/models/unet.py:199-202, through autograd): H, W are the
 * OUTPUT image (dy's), x is the block's input stored at H/2 x W/2; the kernel gathers stored pixel (y >> 1, x >> 1).  ddpm_conv3x3_wgrad_splits
 * is asked with the same (output) H, W. */
int ddpm_conv3x3_wgrad_up_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                               float* dbias, long long bias_stride, int B, int H, int W, int C, int N, int Nreal, int splits,
                               int dtype, void* stream);

/* Weight (and bias) gradient of a 1x1 / stride-1 convolution — autograd of F.conv2d (ddpm_torch/modules.py:120-123) at the attention
 * projections and skip connections (ddpm_torch/models/unet.py:36-37,41 `AttentionBlock.project_in/out/skip`, :80 `ResidualBlock.skip`) — by the slab kernel of csrc/wgrad1x1.hip (bf16):
 *     dW[n][c] = sum_p dy[p][n] * x[p][c]      db[n] = sum_p dy[p][n]      p over the P = B*H*W pixels
 * ddpm_conv1x1_wgrad_splits returns the number of slab copies the kernel writes for this geometry (0: not covered, use
 * ddpm_conv2d_wgrad_nhwc); slice s STORES its partials at dw + s*slab_stride ([N][C] fp32) and dbias + s*bias_stride (dbias may be
 * NULL); ddpm_wgrad_reduce sums the copies in a fixed order. */
int ddpm_conv1x1_wgrad_splits(int P, int C, int N);
int ddpm_conv1x1_wgrad_nhwc(const void* dy, long long dy_ld, const void* x, long long x_ld, float* dw, long long slab_stride,
                            float* dbias, long long bias_stride, int P, int C, int N, int splits, int dtype, void* stream);

int ddpm_wgrad_reduce(const long long* table, int n_tensors, void* stream);

int ddpm_wgrad_unpack(const float* gpack, float* gflat, const long long* descs, int n_tensors, float scale, void* stream);
/* ... the same pass, also producing the sum of squares of everything it writes: the global gradient norm of nn.utils.clip_grad_norm_
 * (ddpm_torch/utils/train.py:159) without another read of all gradients.  total_sq: ddpm_mt_sumsq_slots(n_tensors) floats; every block
 * stores its partial in its own slot behind the 64-float bank and a second launch adds the slots in a FIXED order: on return
 * (stream-ordered) total_sq[0] is the sum and lanes 1..63 are zero — bit-deterministic, so data-parallel replicas holding identical
 * gradients derive the identical clip coefficient (fp32 atomics into the bank, as before round 5, let them drift apart by ulps).
 * total_sq_floats: the capacity of total_sq in floats; DDPM_ERR_SHAPE when it is below ddpm_mt_sumsq_slots(n_tensors) (the buffer was a
 * 64-float bank before round 5: a caller that still passes one is refused instead of being written past). */
int ddpm_wgrad_unpack_sumsq(const float* gpack, float* gflat, const long long* descs, int n_tensors, float scale, float* total_sq,
                            long long total_sq_floats, void* stream);

/* F.linear (modules.py:58-59; unet.py:77,123,125) and torch.einsum in AttentionBlock.qkv (unet.py:46,50):
 *   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k] + bias[n] + residual[b][m][n]   (+ C when accumulate)
 * a_trans/b_trans = 1: the operand is stored [k][m] / [k][n].  out_mode: 0 dtype | 1 fp32 | 2 fp32 atomic add. */
int ddpm_gemm(const void* a, long long a_ld, long long a_bs, int a_trans,
              const void* b, long long b_ld, long long b_bs, int b_trans,
              void* c, long long c_ld, long long c_bs,
              const float* bias, const void* residual, long long res_ld, long long res_bs,
              int M, int N, int K, int batch, float alpha, int accumulate, int out_mode, int splits,
              int dtype, void* stream);

/* Fused single-head attention forward (inference path) — replaces the three products of AttentionBlock.forward
 * (ddpm_torch/models/unet.py:41-52: einsum("bchw,bcHW->bhwHW") * C**-0.5, softmax over the key positions, einsum with v)
 * for a packed projection buffer qkv[B][L][ld] with q at channel 0, k at channel C and v at channel 2C:
 *     out[b][i][:] = sum_j softmax_j(q[b][i] . k[b][j] * scale) v[b][j]        (out pitch out_ld, bf16)
 * The L x L logits stay on chip (LDS / registers).  bf16 only, C in {128, 256}, L a multiple of 128; anything else returns
 * DDPM_ERR_SHAPE and the caller keeps ddpm_gemm + ddpm_softmax_fwd + ddpm_gemm. */
int ddpm_attention_fwd(const void* qkv, long long ld, void* out, long long out_ld, int B, int L, int C, float scale,
                       int dtype, void* stream);

/* Attention for training, and for every geometry ddpm_attention_fwd does not serve (ddpm_torch/models/unet.py:41-52 and its autograd
 * backward): same packed qkv layout.  The forward also writes lse[b][i] = log sum_j exp(q_i . k_j * scale) (fp32, [B][L]; may be NULL
 * for inference); the backward rebuilds the probabilities from q, k and lse on chip — no L x L tensor is ever written — and produces
 *     dV = P^T dO      dS = P o (dO V^T - D),  D[i] = sum_c dO[i][c] O[i][c]      dQ = dS K * scale      dK = dS^T Q * scale
 * into the packed gradient buffer dqkv[B][L][dqkv_ld] (dq at channel 0, dk at C, dv at 2C).  dvec: [B][L] fp32 workspace (receives D).
 * bf16 only; L <= 256 with L % 16 == 0 (the reference attends at 16x16 and below); C <= 512 with C % 32 == 0; else DDPM_ERR_SHAPE. */
int ddpm_attention_fwd_lse(const void* qkv, long long ld, void* out, long long out_ld, float* lse, int B, int L, int C, float scale,
                           int dtype, void* stream);
int ddpm_attention_bwd(const void* qkv, long long ld, const void* o, long long o_ld, const void* d_o, long long do_ld,
                       const float* lse, float* dvec, void* dqkv, long long dqkv_ld, int B, int L, int C, float scale,
                       int dtype, void* stream);

/* nn.GroupNorm(32, C, eps=1e-6) -> SiLU -> Dropout(p) (unet.py:18-20,15,81,85-87,139-140; :57 without SiLU):
 *   y = drop(silu((x - mean_g) * rstd_g * gamma_c + beta_c)),  biased variance over (C/G)*HW elements.
 * stats (optional) receives [B][G][2] = (mean, rstd) for the backward.  workspace: ddpm_gn_workspace_floats().
 * Dropout keep-mask = hash(seed, linear NHWC index) (see ddpm_dropout_mask); scale 1/(1-p).  seed_dev (optional device
 * word): the effective seed is seed + *seed_dev — a captured hipGraph then draws a fresh mask on every replay.
 * Statistics are accumulated about a per-group pivot (the group's first element), so |mean| >> std keeps full accuracy. */
int ddpm_groupnorm_silu_fwd(const void* x, long long x_ld, void* y, long long y_ld, const float* gamma, const float* beta,
                            float* stats, float* workspace, int B, int HW, int C, int G, float eps, int silu,
                            float drop_p, unsigned long long seed, const unsigned long long* seed_dev, int dtype, void* stream);
/* backward of the above: dx (+= when accumulate), dgamma/dbeta += (fp32 atomics).  dx_colsum (optional): receives the per-sample
 * channel sums of dx, dx_colsum[b*colsum_ld + c] = sum_pixels dx[b,:,c] — the gradient of the per-sample time bias that was added
 * in front of this GroupNorm (ddpm_torch/models/unet.py:86); the buffer must be zero on entry (paths without the fused sum add into it).
 * add (optional, NHWC with pitch add_ld): a second gradient contribution added to dx in the same pass — the identity branch of the
 * residual connections (unet.py:60,89): dx = gn_backward(dy) + add  (no separate fan-in add launch). */
int ddpm_groupnorm_silu_bwd(const void* x, long long x_ld, const void* dy, long long dy_ld, void* dx, long long dx_ld,
                            const float* gamma, const float* beta, const float* stats, float* dgamma, float* dbeta,
                            float* workspace, int B, int HW, int C, int G, int silu, float drop_p, unsigned long long seed,
                            const unsigned long long* seed_dev, int accumulate, float* dx_colsum, long long colsum_ld,
                            const void* add, long long add_ld, int dtype, void* stream);
long long ddpm_gn_workspace_floats(int B, int HW, int C, int G, int dtype);

/* get_timestep_embedding (ddpm_torch/functions.py:10-26): out[b] = cat(sin(t_b f), cos(t_b f)) (zero pad if dim odd);
 * freqs[dim/2] = exp(-i ln(1e4)/(dim/2 - 1)) is supplied by the host in fp32. */
int ddpm_timestep_embedding(const long long* t, const float* freqs, float* out, int B, int dim, void* stream);

/* layout / packing helpers (no reference counterpart: the NHWC + packed-weight layout is this library's) */
int ddpm_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int Cp, int dtype, void* stream);
int ddpm_pack_weight(const float* w, void* w_fwd /*[N][R][S][Cp]*/, void* w_dgrad /*[C][R][S][Np], taps flipped*/,
                     int N, int C, int R, int S, int Cp, int Np, int dtype, void* stream);

/* every layer's pack in one launch: descs[i] = {w, w_fwd, w_dgrad (or 0), N, C, R (== S), Cp, Np} as int64 */
int ddpm_pack_weight_multi(const long long* descs, int n_tensors, int dtype, void* stream);

/* GaussianDiffusion.q_sample (ddpm_torch/diffusion.py:92-97): xt = sqrt_ab[t]*x0 + sqrt_1mab[t]*noise  (fp32, [B][n]).
 * T = table length: a sample whose t is outside [0, T) is filled with NaN (the reference's gather raises, :83). */
int ddpm_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ab, const float* sqrt_1mab,
                  float* xt, int B, int n, int T, void* stream);
/* flat_mean((target - pred)^2) (diffusion.py:239, functions.py:99-101) and its gradient w.r.t. pred */
int ddpm_mse_fwd(const float* pred, const float* target, float* loss, int B, int n, void* stream);
int ddpm_mse_bwd(const float* pred, const float* target, const float* gloss, float* gpred, int B, int n, void* stream);
/* out[0] = sum_i x[i]*w[i], one block, fixed order: the batch mean of the per-sample losses (utils/train.py:151 `loss.mean()`) */
int ddpm_weighted_sum_f32(const float* x, const float* w, float* out, int n, void* stream);
/* C[m][n] = sum_k A[k][m] * B[k][n] in fp32 for short reductions (K = the batch): the weight gradients of the time-embedding path —
 * autograd of F.linear (ddpm_torch/modules.py:58-59) at ResidualBlock.fc (models/unet.py:77,86) and UNet.embed (:122-126): dW = d(out)^T in.
 * A is [K][lda], B is [K][ldb], C is [M][ldc]; plain stores in a fixed order (bit-deterministic). */
int ddpm_atb_f32(const float* a, long long lda, const float* b, long long ldb, float* c, long long ldc, int M, int N, int K, void* stream);
/* p_mean_var + p_sample_step (diffusion.py:107-158; ddim.py inherits it): one fused update
 *   x0 = clamp(recip[t]*x_t - recip_m1[t]*out)   (mean_type 0 = eps; 1: x0 = out; 2: out is the mean)
 *   x_prev = coef1[t]*x0 + coef2[t]*x_t + 1[t>0]*exp(0.5*logvar[t])*z ;  pred_x0 optional.
 * The clamp propagates NaN like torch.clamp; T = table length, t outside [0, T) poisons the sample with NaN. */
int ddpm_p_sample_step(const float* x_t, const float* model_out, const float* z, const long long* t,
                       const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                       const float* logvar, float* x_prev, float* pred_x0, int B, int n, int mean_type, int clip, int T, void* stream);
/* Variational-bound term L_t in bits per dimension — ddpm_torch/diffusion.py:203-215 `_loss_term_bpd` (loss_type = "kl", :222-224;
 * `calc_all_bpd`, :251-267): per sample b the mean over the n elements of KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) / ln 2 for t[b] > 0
 * (functions.py:30-36 `normal_kl`, both log-variances table entries) or of the discretized-Gaussian decoder NLL / ln 2 for t[b] = 0
 * (functions.py:39-65).  The model mean is derived from the network output as in ddpm_p_sample_step (mean_type 0 eps | 1 x_0 | 2 mean;
 * clip as `clip_denoised`); pred_x0 (may be null) receives the x_0 estimate.  Tables: fp32, length T; t outside [0, T) poisons the row.
 * ddpm_vlb_terms_bwd: gout[b][i] = gloss[b] * d loss[b] / d model_out[b][i] (clip_denoised = False, as `train_losses` calls it). */
int ddpm_vlb_terms(const float* x_0, const float* x_t, const float* model_out, const long long* t,
                   const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                   const float* post_logvar, const float* model_logvar, float* loss, float* pred_x0,
                   int B, int n, int mean_type, int clip, int T, void* stream);
int ddpm_vlb_terms_bwd(const float* x_0, const float* x_t, const float* model_out, const long long* t,
                       const float* sqrt_recip_ab, const float* sqrt_recip_m1_ab, const float* post_coef1, const float* post_coef2,
                       const float* post_logvar, const float* model_logvar, const float* gloss, float* gout,
                       int B, int n, int mean_type, int T, void* stream);
/* subsequence.gather(0, t) (ddim.py:101) and t += delta (t.fill_ in diffusion.py:172, device-side for graph replay) */
int ddpm_gather_i64(const long long* idx, const long long* map, long long* out, int B, void* stream);
int ddpm_add_i64(long long* t, int B, long long delta, void* stream);
/* out[r][:] = table[idx[r]][:], fp32 rows of row_len (% 4 == 0) floats; idx outside [0, table_rows) -> the row is NaN.  Sampling only: every
 * ResidualBlock's time bias fc(act(t_emb)) (unet.py:86, with UNet.embed :122-126,207 in front of it) depends on t alone, so the sampler
 * precomputes the [T][sum Cout] table once per weight version and each step gathers its rows instead of running the embedding MLP. */
int ddpm_gather_rows_f32(const float* table, const long long* idx, float* out, int rows, int row_len, int table_rows, void* stream);

/* nn.SiLU on the time-embedding path (unet.py:86,124) */
int ddpm_silu_fwd(const float* x, float* y, long long n, void* stream);
int ddpm_silu_bwd(const float* x, const float* dy, float* dx, long long n, int accumulate, void* stream);

/* bias / time-bias gradients: per_sample[b][c] += sum_pixels dy, total[c] += sum_{b,pixels} dy (fp32 atomics into
 * zero-initialised buffers); C <= 64 x 256 16-byte vectors per launch (groups of 256 on blockIdx.z) */
int ddpm_colsum(const void* dy, long long ld, float* per_sample, long long ps_ld, float* total, int B, int HW, int C, int dtype, void* stream);
/* backward of nn.Upsample(2,"nearest"): dx[b,y,x,c] (+)= sum of the 2x2 block of dy_up */
int ddpm_upsample2x_bwd(const void* dy_up, void* dx, long long dx_ld, int B, int H, int W, int C, int accumulate, int dtype, void* stream);
/* 2x resampling WITHOUT a convolution — UNet(resample_with_conv=False): ddpm_torch/models/unet.py:169 `nn.AvgPool2d(2)`, :196
 * `nn.Upsample(scale_factor=2, mode="nearest")` on its own — and their autograd.  NHWC with pixel pitches; H, W = the SMALL grid.
 *   up = 0: y[b,h,w,:] (+)= scale * (x[b,2h,2w,:] + x[b,2h,2w+1,:] + x[b,2h+1,2w,:] + x[b,2h+1,2w+1,:])   (AvgPool forward: 0.25; Upsample backward: 1)
 *   up = 1: y[b,2h+a,2w+b,:] (+)= scale * x[b,h,w,:], a, b in {0, 1}                                       (Upsample forward: 1; AvgPool backward: 0.25) */
int ddpm_resample2x_nhwc(const void* x, long long x_ld, void* y, long long y_ld, int B, int H, int W, int C, int up, float scale,
                         int accumulate, int dtype, void* stream);
/* y (+)= x over [rows][C] slices with pitches (gradient fan-in of the residual / skip connections) */
int ddpm_add_rows(const void* x, long long x_ld, void* y, long long y_ld, long long rows, int C, int accumulate, int dtype, void* stream);

/* torch.softmax over the keys (unet.py:47-49) and its backward dS = P*(dP - sum(dP*P)); rows of length L */
int ddpm_softmax_fwd(const float* s, void* p, long long rows, int L, int dtype, void* stream);
int ddpm_softmax_bwd(const void* p, const float* dp, void* ds, long long rows, int L, int dtype, void* stream);

/* nn.utils.clip_grad_norm_ + torch.optim.Adam + EMA.update (ddpm_torch/utils/train.py:159-165,300-305; train.py:128)
 * over one fp32 tensor: total_sq[0] += ||g||^2 (workspace >= 1024 floats); then one fused pass
 *   g' = g * min(1, max_norm/(sqrt(total_sq)+1e-6)); m,v <- Adam moments; p -= lr/bc1 * m/(sqrt(v/bc2)+eps);
 *   shadow += ema_w * (p - shadow)      (total_sq null or max_norm <= 0: no clipping; shadow null: no EMA). */
int ddpm_sumsq_accumulate(const float* g, long long n, float* total_sq, float* workspace, void* stream);
int ddpm_adam_ema_step(float* p, const float* g, float* m, float* v, float* shadow, long long n, const float* total_sq,
                       float max_norm, float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                       float ema_w, void* stream);

/* multi-tensor forms: ONE launch over every parameter tensor.  table[i] = {p, g, m, v, shadow (0 = none), numel} (int64).
 * ddpm_mt_grad_sumsq: total_sq holds ddpm_mt_sumsq_slots(n_tensors) floats; on return total_sq[0] = sum ||g_i||^2 (lanes 1..63 zero),
 * added up in a fixed order (see ddpm_wgrad_unpack_sumsq; total_sq_floats < ddpm_mt_sumsq_slots(n_tensors) -> DDPM_ERR_SHAPE).
 * ddpm_mt_adam_ema: the fused update above for all i, with the clipping norm taken from the sum of the 64-float bank at total_sq — it
 * reads total_sq[0..63] and nothing behind them, so a 64-float buffer is enough for this call. */
int ddpm_mt_sumsq_slots(int n_tensors);
int ddpm_mt_grad_sumsq(const long long* table, int n_tensors, float* total_sq, long long total_sq_floats, void* stream);
/* hyper_dev (optional, 4 device floats {lr, bias_corr1, bias_corr2, ema_w}) overrides the by-value scalars: the values of the
 * current step are read from memory, so one captured hipGraph serves every training step. */
int ddpm_mt_adam_ema(const long long* table, int n_tensors, const float* total_sq, float max_norm, float lr, float beta1,
                     float beta2, float eps, float bias_corr1, float bias_corr2, float ema_w, const float* hyper_dev, void* stream);

/* dst_i = a_i (+ b_i) over many small fp32 tensors in one launch: table[i] = {a, b (0 = none), dst, numel} (int64).  Builds the
 * concatenated time-bias projection (all ResidualBlock.fc weights; fc.bias + conv1.bias, ddpm_torch/models/unet.py:77,85-86). */
int ddpm_mt_gather_f32(const long long* table, int n_tensors, void* stream);

/* ---- launch plans (no upstream counterpart: the reference issues its step op by op from Python, ddpm_torch/utils/train.py:148-170).
 * The training step is a fixed sequence of the calls above; a plan records it once — entry-point name + argument words + the
 * hipStream_t each call was given — and ddpm_plan_run re-issues a whole segment from C: eager semantics (two concurrent streams,
 * host work such as RCCL calls between segments) without the interpreter between launches.  Step-varying scalars must live in
 * device memory (hyper_dev, seed_dev), exactly as for a captured hipGraph; every pointer recorded must stay valid while the plan lives.
 *
 * ddpm_stream_order(waiter, signaller): everything enqueued on hipStream_t `waiter` after the call runs after everything enqueued on
 * `signaller` before it (event record + stream wait on a pooled event) — the fork / join edges of the weight-gradient stream.
 * ddpm_fill_zero: stream-ordered memset(0) (torch.zeros / Tensor.zero_ of the gradient staging buffers).
 * ddpm_plan_append: words[i] = argument i of `entry` as a 64-bit word (pointers / 64-bit integers as they are, int in the low half,
 * float as its IEEE-754 bits in the low half); returns the entry index or -(status) (unknown name or wrong argument count: -1).
 * ddpm_plan_cut closes the current segment (returns its index).  ddpm_plan_run issues segment `segment` and returns 0 or the status of
 * the first failing call (ddpm_plan_failed_entry names it).  A plan is not thread-safe; distinct plans are independent. */
int ddpm_stream_order(void* waiter, void* signaller);
int ddpm_fill_zero(void* p, long long bytes, void* stream);
void* ddpm_plan_create(void);
int ddpm_plan_destroy(void* plan);
int ddpm_plan_append(void* plan, const char* entry, const unsigned long long* words, int n_words);
int ddpm_plan_cut(void* plan);
int ddpm_plan_segments(void* plan);
int ddpm_plan_entries(void* plan);
int ddpm_plan_run(void* plan, int segment);
const char* ddpm_plan_failed_entry(void* plan, int* index);
int ddpm_plan_entry_arity(const char* entry);

#ifdef __cplusplus
}
#endif
#endif
