// Activation storage types.  Arithmetic is always fp32; BF16 = true means the activation tensors
// (inputs, residual, output) are stored as bfloat16 in HBM: widened on load (a 16-bit shift, exact),
// rounded to nearest-even on store.  Parameters and packed weights stay fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace unetmi {

// A kernel attribute (dynamic LDS beyond 64 KB) has to be raised once per kernel AND per device.  One of these per
// kernel instantiation: a bit per device ordinal, set after the attribute call succeeded on that device.  Two
// threads racing on the first launch both make the (idempotent) call; nobody launches before it has succeeded.
struct PerDeviceLatch {
    std::atomic<unsigned long long> done[4] = {};      // 256 device ordinals
    template <class F>
    hipError_t once(F&& raise) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if ((unsigned)dev >= 256u) return raise();
        std::atomic<unsigned long long>& word = done[dev >> 6];
        const unsigned long long bit = 1ull << (dev & 63);
        if (word.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = raise();
        if (e == hipSuccess) word.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned)b << 16);
}

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                                      // round to nearest even
    return (unsigned short)(u >> 16);
}

// Transcendental activations on the hardware's own exp / log / reciprocal instructions (each accurate to about one
// unit in the last place) instead of the math library's sequences, which are 30-60 instructions each: per value that
// is more work than a thin layer's multiply-adds, and in the recurrence kernels it sits on the critical path of every
// time step.  Absolute error <= ~2e-7 (tanh and softplus lose RELATIVE accuracy only where their value is ~1e-7).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __expf(-2.f * fabsf(x));
    return copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
}
__device__ __forceinline__ float fast_softplus(float x) {          // log(1 + e^x) = max(x, 0) + log1p(e^-|x|)
    const float t = __expf(-fabsf(x));
    return fmaxf(x, 0.f) + (t < 2e-4f ? t * (1.f - 0.5f * t) : __logf(1.f + t));
}

template <bool BF16>
__device__ __forceinline__ float ld_act(const void* p, size_t i) {
    if (BF16) return bf16_bits_to_f32(static_cast<const unsigned short*>(p)[i]);
    return static_cast<const float*>(p)[i];
}

template <bool BF16>
__device__ __forceinline__ void st_act(void* p, size_t i, float v) {
    if (BF16) static_cast<unsigned short*>(p)[i] = f32_to_bf16_bits(v);
    else static_cast<float*>(p)[i] = v;
}

// buffer-descriptor forms: byte offsets scale with the element size; out-of-range lanes read 0 / drop the store
template <bool BF16>
__device__ __forceinline__ float buf_ld_act(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if (BF16) return bf16_bits_to_f32((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

template <bool BF16>
__device__ __forceinline__ void buf_st_act(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if (BF16) __builtin_amdgcn_raw_buffer_store_b16((short)f32_to_bf16_bits(v), rs, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, voff, soff, 0);
}

// four consecutive activations per lane: one 16-byte (fp32) or 8-byte (bf16) access; needs only element alignment
typedef float io_f32x4 __attribute__((ext_vector_type(4)));
typedef int io_i32x4 __attribute__((ext_vector_type(4)));
typedef int io_i32x2 __attribute__((ext_vector_type(2)));

template <bool BF16>
__device__ __forceinline__ io_f32x4 buf_ld_act4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if (BF16) {
        const io_i32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        return io_f32x4{__builtin_bit_cast(float, (unsigned)w[0] << 16), __builtin_bit_cast(float, (unsigned)w[0] & 0xffff0000u),
                        __builtin_bit_cast(float, (unsigned)w[1] << 16), __builtin_bit_cast(float, (unsigned)w[1] & 0xffff0000u)};
    }
    return __builtin_bit_cast(io_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

template <bool BF16>
__device__ __forceinline__ void buf_st_act4(io_f32x4 v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if (BF16) {
        io_i32x2 w;
        w[0] = (int)((unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16));
        w[1] = (int)((unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16));
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, voff, soff, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(io_i32x4, v), rs, voff, soff, 0);
    }
}

template <bool BF16> constexpr int act_bytes() { return BF16 ? 2 : 4; }

// Lanes [32, 64) of a and lanes [0, 32) of b change places (v_permlane32_swap_b32).  Written as inline assembly, with
// the wait states around it that the compiler puts around its own: in some uses of __builtin_amdgcn_permlane32_swap
// this compiler took the instruction's two results for one value (it added them as 2 * first, or stored the first
// one twice and reused the second one's register).
__device__ __forceinline__ void swap_lane_halves(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

}  // namespace unetmi
