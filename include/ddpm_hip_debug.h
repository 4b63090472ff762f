/* ddpm_hip_debug.h — instrumentation, measurement and test hooks of libddpm_hip.so.
 *
 * NOT part of the interface a maintainer binds to replace the reference's ATen calls (that is ddpm_hip.h): nothing here has an upstream
 * counterpart, nothing here is needed to train or to sample.  bench.py uses the dispatch queries and the probes to attribute its
 * per-launch timings and to measure what the chip sustains; tests/ uses the mask hook and the fault record; the data-parallel study
 * (scripts/dp_one_rank.py) uses the reserved-CU switch and the copy probe.  Same conventions as ddpm_hip.h (device pointers, status codes). */
#ifndef DDPM_HIP_DEBUG_H
#define DDPM_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* Instrumentation (no upstream counterpart): which kernel ddpm_conv2d_nhwc / ddpm_conv2d_wgrad_nhwc / ddpm_gemm would dispatch a
 * call with these arguments to — 1 gemm_kernel (4 waves), 2 gemm_kernel (8 waves), 3 gemm_kernel (deep LDS ring), 4 gemm64_kernel
 * (64x64 tiles), 5 conv3x3_halo_kernel, 6 wgrad3x3 (patch-stationary / wave-specialised), 7 pw_conv_kernel (persistent streaming 1x1 conv),
 * 8 / 10 conv3x3_stream_kernel (persistent stationary-halo 3x3 conv, 16x16 / 8x8 patches), 9 wgrad1x1_kernel, 11 / 12 conv3x3_few_out / few_in (edge convs),
 * 13 conv3x3_pc_kernel (wave-specialised 3x3 conv); a negative value is -(status code) for arguments the launching call would reject.
 * Pure functions of their arguments: the same dispatch code runs with launching switched off, nothing is retained between calls.
 * bench.py uses them to attribute its per-launch HIP-event timings to the kernel that ran. */
int ddpm_conv2d_variant(long long x_ld, long long y_ld, int B, int H, int W, int C, int Ho, int Wo, int N, int R, int S,
                        int stride, int pad_t, int pad_l, int upsample, int dilate, int out_mode, int splits, int dtype, int epilogue);
int ddpm_conv2d_wgrad_variant(long long dy_ld, long long x_ld, int B, int H, int W, int C, int Creal, int Ho, int Wo, int N, int Nreal,
                              int R, int S, int stride, int pad_t, int pad_l, int upsample, int splits, int dtype);
int ddpm_gemm_variant(long long a_ld, int a_trans, long long b_ld, int b_trans, long long c_ld, int M, int N, int K, int batch,
                      int out_mode, int splits, int dtype);
/* ... and which kernel ddpm_conv3x3_wgrad_nhwc / _up_nhwc runs for a geometry: 14 wgrad3x3_ws_kernel (wave-specialised, 64 x 64 x 9 tiles:
 * 16-divisible images, C % 64 == 0, N % 64 == 0), 6 wgrad3x3_kernel (patch-stationary, 64 x 32 x 9), -1 not covered. */
int ddpm_conv3x3_wgrad_variant(int B, int H, int W, int C, int N);

/* measurement hook (bench.py's roofline leg; no upstream counterpart): one launch of an MFMA-only loop on every CU — 256 blocks x 4 waves,
 * 16 * iters v_mfma_f32_32x32x16_bf16 per wave on register-resident random bf16 operands (zero_operands = 1: zeros) =
 * iters * 16 * 32768 * 1024 FLOP.  Timed over a few hundred ms it gives the rate the matrix pipe SUSTAINS at the chip's power
 * budget (~1.7 PFLOP/s on random operands, ~2.5 on zeros).  sink: >= 256 floats, never written in practice. */
int ddpm_mfma_probe(float* sink, int iters, int zero_operands, void* stream);

/* diagnostic (no upstream counterpart): the wave-specialised 3x3 kernel's LDS-semaphore waits are bounded in wall time (~5 s); one that
 * expires records {block, counter address, value waited for, kind} and traps (the launch fails instead of hanging the GPU).  out4 is HOST
 * memory; kind 0 = no wait has ever expired.  Setting DDPM_CONV_NO_PC=1 runs those calls on the barrier-synchronised kernel instead. */
int ddpm_conv3x3_pc_last_fault(unsigned* out4);
/* ... the same record for the wave-specialised 3x3 weight-gradient kernel (DDPM_WGRAD3_NO_WS=1 runs those calls on the barrier-synchronised one). */
int ddpm_wgrad3x3_ws_last_fault(unsigned* out4);

/* Data-parallel training (upstream: DistributedDataParallel's all-reduce beside the backward, train.py:110): the persistent kernels size
 * their grids to the whole chip, one block per compute unit, so a collective's kernel issued from inside the backward finds a CU only at
 * a block boundary.  ddpm_set_reserved_cus(n) makes every persistent launcher (3x3 / 1x1 conv, both weight-gradient kernels) plan for
 * 256 - n compute units (0 <= n <= 192; process-wide; 0 = default).  Changes the slab counts ddpm_conv3x3_wgrad_splits /
 * ddpm_conv1x1_wgrad_splits report: set it before asking.  ddpm_copy_probe: `blocks` workgroups streaming `bytes` (a multiple of 16)
 * from src to dst — a stand-in for a ring step's copy kernel, used by bench.py to measure what a collective gets on one GPU. */
int ddpm_set_reserved_cus(int n);
int ddpm_get_reserved_cus(void);
int ddpm_copy_probe(void* dst, const void* src, long long bytes, int blocks, void* stream);

/* test hook: the keep-mask (1/0) the GroupNorm kernels regenerate for element indices 0..n-1 */
int ddpm_dropout_mask(float* mask, long long n, float p, unsigned long long seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
