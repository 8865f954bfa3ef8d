// optimizer.hip - mixed-precision Adam with global-norm gradient clipping over a LIST of tensors, two launches per step
// (three above 16 k chunks = 33 M parameters: see adam_norm_reduce_k).
//
// Reference: the optimizer the recipe builds at megatron/training.py:571-580 (apex / DeepSpeed FusedAdam, adam_w_mode,
// betas (0.9, 0.95), eps 1e-8: configs/magma_setup.yml:38-45) under DeepSpeed's 16-bit wrapper, i.e. fp32 master weights
// and fp32 moments, 16-bit model weights, and `gradient_clipping: 1.0` on the global L2 norm
// (configs/magma_pythia_410M.yml:57).  Those libraries are not part of the reference tree (parity unpinned);
// this file states their published update rule:
//     g' = g * min(1, max_norm / (||g||_2 + 1e-6))
//     m = b1 m + (1 - b1) g',  v = b2 v + (1 - b2) g'^2
//     p = p - lr * ( (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps) + wd * p )          (decoupled decay, adam_w_mode)
//     w16 = round(p)
// HBM-bound: 2 B (g) + 12 B read + 12 B + 2 B written per parameter; the trainable set of the 410M recipe (13.7 M
// parameters in ~200 tensors) is one launch of ~6.7 k blocks instead of a few hundred small ones.
// Determinism: per-block partial sums of g^2 are written to a workspace and every block of the update kernel re-adds
// them in the same fixed order (no float atomics).
#include "common.h"

namespace mm {

constexpr int OPT_CHUNK = 2048;  // elements per block: 256 threads x 8

struct AdamChunk {  // one block's work; the table is built once per optimizer (shapes are static)
    int tensor;     // index into the per-step pointer tables
    int count;      // valid elements in this chunk (<= OPT_CHUNK)
    long long toff; // element offset inside the tensor
    long long foff; // element offset inside the flat fp32 state buffers
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void adam_sqnorm_k(float* __restrict__ partial, int* __restrict__ step,
                                                     const AdamChunk* __restrict__ chunks, const u16* const* __restrict__ grads) {
    __shared__ float red[4];
    const AdamChunk c = chunks[blockIdx.x];
    const u16* g = grads[c.tensor];
    float acc = 0.f;
    const int i = threadIdx.x * 8;
    if (g != nullptr && i < c.count) {
        if (i + 8 <= c.count && ((((uintptr_t)(g + c.toff + i)) & 15) == 0)) {
            float f[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(g + c.toff + i), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += f[j] * f[j];
        } else {
            for (int j = i; j < min(i + 8, c.count); ++j) {
                const float f = to_f32<T>(g[c.toff + j]);
                acc += f * f;
            }
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s;
        if (blockIdx.x == 0) *step += 1;  // nobody reads it in this launch; the update kernel sees the new value
    }
}

// Large models: with hundreds of thousands of chunks "every block re-adds every partial" is quadratic (1.3 B parameters: 700 k
// blocks x 700 k loads - measured 29 ms for a 360 M-parameter model where the update itself takes 3).  Above OPT_READD_MAX
// chunks one block sums the partials once, in a fixed order (thread-strided, then the block tree: deterministic), and leaves
// the total in partial[0] for the update kernel.
constexpr int OPT_READD_MAX = 16384;

__global__ __launch_bounds__(1024) void adam_norm_reduce_k(float* __restrict__ partial, int nchunks) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 1024) acc += partial[i];
    acc = wave_sum(acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = acc;
    __syncthreads();  // (also: every thread has read its partials before partial[0] is overwritten)
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        partial[0] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adam_update_k(u16* const* __restrict__ params, float* __restrict__ master,
                                                     float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                     float* __restrict__ grad_norm_out, const float* __restrict__ partial,
                                                     int nchunks, const int* __restrict__ step, const AdamChunk* __restrict__ chunks,
                                                     const u16* const* __restrict__ grads, float lr, const float* __restrict__ lr_dev,
                                                     float b1, float b2, float eps, float wd, float max_norm, int bias_correction,
                                                     int pre_reduced) {
    __shared__ float red[4];
    __shared__ float sh[4];
    // global norm: every block adds the same partials in the same order (or reads the total adam_norm_reduce_k left)
    float acc = 0.f;
    if (pre_reduced) {
        if (threadIdx.x == 0) acc = partial[0];
    } else {
        for (int i = threadIdx.x; i < nchunks; i += 256) acc += partial[i];
    }
    const float total = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(total);
        float clip = 1.0f;
        if (max_norm > 0.f) clip = fminf(1.0f, max_norm / (norm + 1e-6f));
        const int t = *step;
        double bc1 = 1.0, bc2 = 1.0;
        if (bias_correction) {
            bc1 = 1.0 - pow((double)b1, (double)t);
            bc2 = 1.0 - pow((double)b2, (double)t);
        }
        sh[0] = clip;
        sh[1] = (float)(1.0 / bc1);
        sh[2] = (float)(1.0 / sqrt(bc2));
        sh[3] = lr_dev ? *lr_dev : lr;  // device-resident learning rate: a scheduler's value reaches a replayed HIP graph
        if (blockIdx.x == 0 && grad_norm_out) *grad_norm_out = norm;
    }
    __syncthreads();
    const float clip = sh[0], inv_bc1 = sh[1], inv_sqrt_bc2 = sh[2];
    lr = sh[3];
    const AdamChunk c = chunks[blockIdx.x];
    const u16* g = grads[c.tensor];
    if (g == nullptr) return;  // a parameter without a gradient this step is left alone (torch / apex semantics)
    u16* w = params[c.tensor];
    const int i0 = threadIdx.x * 8;
    if (i0 >= c.count) return;
    const int n = min(8, c.count - i0);
    const bool vec = n == 8 && ((((uintptr_t)(g + c.toff + i0)) | ((uintptr_t)(w + c.toff + i0))) & 15) == 0;
    float gf[8], p[8], m[8], v[8];
    if (vec) {
        unpack8<T>(*reinterpret_cast<const u32x4*>(g + c.toff + i0), gf);
    } else {
        for (int j = 0; j < n; ++j) gf[j] = to_f32<T>(g[c.toff + i0 + j]);
    }
    const long long fo = c.foff + i0;  // chunks start at multiples of 8 elements of the flat buffers: 32-byte aligned
    if (n == 8) {
        *reinterpret_cast<f32x4*>(p) = *reinterpret_cast<const f32x4*>(master + fo);
        *reinterpret_cast<f32x4*>(p + 4) = *reinterpret_cast<const f32x4*>(master + fo + 4);
        *reinterpret_cast<f32x4*>(m) = *reinterpret_cast<const f32x4*>(exp_avg + fo);
        *reinterpret_cast<f32x4*>(m + 4) = *reinterpret_cast<const f32x4*>(exp_avg + fo + 4);
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(exp_avg_sq + fo);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(exp_avg_sq + fo + 4);
    } else {
        for (int j = 0; j < n; ++j) {
            p[j] = master[fo + j];
            m[j] = exp_avg[fo + j];
            v[j] = exp_avg_sq[fo + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < n) {
            const float gj = gf[j] * clip;
            m[j] = b1 * m[j] + (1.0f - b1) * gj;
            v[j] = b2 * v[j] + (1.0f - b2) * gj * gj;
            const float denom = sqrtf(v[j]) * inv_sqrt_bc2 + eps;
            p[j] = p[j] - lr * ((m[j] * inv_bc1) / denom + wd * p[j]);
        }
    }
    if (n == 8) {
        *reinterpret_cast<f32x4*>(master + fo) = *reinterpret_cast<const f32x4*>(p);
        *reinterpret_cast<f32x4*>(master + fo + 4) = *reinterpret_cast<const f32x4*>(p + 4);
        *reinterpret_cast<f32x4*>(exp_avg + fo) = *reinterpret_cast<const f32x4*>(m);
        *reinterpret_cast<f32x4*>(exp_avg + fo + 4) = *reinterpret_cast<const f32x4*>(m + 4);
        *reinterpret_cast<f32x4*>(exp_avg_sq + fo) = *reinterpret_cast<const f32x4*>(v);
        *reinterpret_cast<f32x4*>(exp_avg_sq + fo + 4) = *reinterpret_cast<const f32x4*>(v + 4);
    } else {
        for (int j = 0; j < n; ++j) {
            master[fo + j] = p[j];
            exp_avg[fo + j] = m[j];
            exp_avg_sq[fo + j] = v[j];
        }
    }
    if (vec) {
        *reinterpret_cast<u32x4*>(w + c.toff + i0) = pack8<T>(p);
    } else {
        for (int j = 0; j < n; ++j) w[c.toff + i0 + j] = from_f32<T>(p[j]);
    }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_adam_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int mm_adam_step(void* const* params, const void* const* grads, float* master, float* exp_avg, float* exp_avg_sq,
                            const void* chunks, int nchunks, float* partial, int* step, float* grad_norm_out, float lr,
                            const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, int bias_correction,
                            int dtype, void* stream) {
    MM_CHECK_ARG(params && grads && master && exp_avg && exp_avg_sq && chunks && partial && step, "mm_adam_step: null pointer");
    MM_CHECK_ARG(nchunks > 0, "mm_adam_step: nchunks=%d must be positive", nchunks);
    MM_CHECK_ARG(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f,
                 "mm_adam_step: hyper-parameters out of range (lr=%g betas=(%g, %g) eps=%g wd=%g)", (double)lr, (double)beta1,
                 (double)beta2, (double)eps, (double)weight_decay);
    MM_CHECK_ARG(aligned16(master) && aligned16(exp_avg) && aligned16(exp_avg_sq), "mm_adam_step: state buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        adam_sqnorm_k<T><<<nchunks, 256, 0, st>>>(partial, step, (const AdamChunk*)chunks, (const u16* const*)grads);
        MM_CHECK_LAUNCH();
        const int pre_reduced = nchunks > OPT_READD_MAX;
        if (pre_reduced) {
            adam_norm_reduce_k<<<1, 1024, 0, st>>>(partial, nchunks);
            MM_CHECK_LAUNCH();
        }
        adam_update_k<T><<<nchunks, 256, 0, st>>>((u16* const*)params, master, exp_avg, exp_avg_sq, grad_norm_out, partial, nchunks,
                                                  step, (const AdamChunk*)chunks, (const u16* const*)grads, lr, lr_dev, beta1, beta2, eps,
                                                  weight_decay, max_grad_norm, bias_correction, pre_reduced);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}
