/*
 * behavenet_hip_debug.h -- C ABI of tests/native/libbn_debug.so: TEST-ONLY hardware probes and
 * the LDS-poisoning aid of the GPU parity tests.  Not part of the product library
 * (libbehavenet_hip.so does not contain or need any of these); built by
 * `make -C tests/native` (and by __graft_entry__.build()).
 *
 * Same conventions as behavenet_hip.h: device pointers, caller-owned buffers, the caller's
 * hipStream_t passed as void*, 0 on success / positive hipError_t on a failed launch.
 */
#ifndef BEHAVENET_HIP_DEBUG_H
#define BEHAVENET_HIP_DEBUG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Fill (almost all of) every CU's LDS with quiet NaNs: a kernel that afterwards reads LDS it
 * never wrote (a padded tap with a zero weight, a skipped halo) produces NaN instead of passing
 * on a previous kernel's leftovers.  `sink`: one device float (keeps the stores alive). */
int bn_debug_poison_lds(float* sink, void* stream);

/* MFMA issue-rate ceilings: register operands / operands read from LDS in the conv kernels'
 * shapes ("Why the matrix cores"). */
int bn_debug_probe_mfma(float* out, int blocks, int iters, void* stream);
int bn_debug_probe_mfma_lds(float* out, int blocks, int threads, int iters, int mode,
                            void* stream);

/* HBM store-stream ceilings (plain fill, fill variants, enc.conv0-shaped store streams). */
int bn_debug_probe_fill(float* out, size_t n, int blocks, void* stream);
int bn_debug_probe_fill2(float* out, size_t n, int blocks, int mode, void* stream);
int bn_debug_probe_fill3(float* out, int n_frames, void* stream);
int bn_debug_probe_fill4(float* out, int n_frames, int mode, int grid, void* stream);

/* Semantics of an out-of-range lane in `buffer_load ... lds` (LDS keeps its old value). */
int bn_debug_probe_lds_dma(const float* p, float* o, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
