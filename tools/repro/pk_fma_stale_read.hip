// Stand-alone probe for the stale read DESIGN.md 4.4d item 8 describes (gfx950): a float32 VALU result consumed by the
// packed-float32 instruction issued directly behind it as a broadcast (op_sel_hi) source.
//   variant 0: v_mul_f32 t, k, b ; v_pk_fma_f32 acc, x, t(broadcast), acc          (adjacent)
//   variant 1: v_mul_f32 t, k, b ; v_mov_b32 (unrelated) ; v_pk_fma_f32 ...        (one VALU instruction in between)
//   variant 2: as 0, behind an MFMA (the accumulation kernel's neighbourhood)
// Every lane accumulates acc += x * (k * b) for `iters` different (x, k, b) and the host recomputes the same sums with
// scalar float32 FMAs (fmaf): any difference is a wrong read.  hipcc --offload-arch=gfx950 -O2 pk_fma_stale_read.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// The accumulation kernel's own sequence, register for register (investigation build GVAR = 1), with the
// MFMA whose B operand is v[80:83] in front; ORDER 1 = the failing build's order, ORDER 6 = the passing one's (the second
// v_mul_f32 moved up).  acc = v[68:71] += x0 * (k b0) + x1 * (k b1) for x0 = v[76:79], x1 = v[72:75].
template <int ORDER>
__global__ __launch_bounds__(256, 4) void probe_seq(int iters, const float *__restrict__ in, float *__restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f16v c = {0};
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < iters; ++i) {
        const float k = in[(size_t)(3 * i + 0) * 64 + (threadIdx.x & 63)];
        const float b = in[(size_t)(3 * i + 1) * 64 + (threadIdx.x & 63)];
        const float xv = in[(size_t)(3 * i + 2) * 64 + (threadIdx.x & 63)];
        f4 x0 = {xv, xv + 1.f, xv + 2.f, xv + 3.f}, x1 = {xv + 4.f, xv + 5.f, xv + 6.f, xv + 7.f};
        const float b1 = b + 0.5f;
        if (ORDER == 1)
            asm volatile(
                "v_mov_b32 v80, 1.0\n\tv_mov_b32 v81, 1.0\n\tv_mov_b32 v82, 1.0\n\tv_mov_b32 v83, 1.0\n\t"
                "v_mov_b32 v112, %[b0]\n\tv_mov_b32 v113, %[b1]\n\tv_mov_b32 v110, %[k]\n\t"
                "s_nop 4\n\t"
                "v_mfma_f32_32x32x16_f16 %[c], %[a], v[80:83], %[c]\n\t"
                "v_mov_b32 v80, %[k]\n\t"
                "v_mul_f32 v66, v80, v112\n\t"
                "s_mulk_i32 s2, 0x4a00\n\t"
                "v_pk_fma_f32 %[acc01], %[x0a], v[66:67], %[acc01] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[acc23], %[x0b], v[66:67], %[acc23] op_sel_hi:[1,0,1]\n\t"
                "v_fma_mixlo_f16 v66, %[k], v110, 0\n\t"
                "v_add_u32 v81, s2, v81\n\t"
                "v_mul_f32 v80, v80, v113\n\t"
                "v_fma_mixhi_f16 v66, %[b0], v110, 0\n\t"
                "v_pk_fma_f32 %[acc01], %[x1a], v[80:81], %[acc01] op_sel_hi:[1,0,1]\n\t"
                "v_fma_mixlo_f16 v67, %[k], v110, -v66 op_sel_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %[acc23], %[x1b], v[80:81], %[acc23] op_sel_hi:[1,0,1]"
                : [acc01] "+v"(*(f2 *)&acc), [acc23] "+v"(*((f2 *)&acc + 1)), [c] "+v"(c)
                : [a] "v"(a), [k] "v"(k), [b0] "v"(b), [b1] "v"(b1), [x0a] "v"(*(f2 *)&x0), [x0b] "v"(*((f2 *)&x0 + 1)),
                  [x1a] "v"(*(f2 *)&x1), [x1b] "v"(*((f2 *)&x1 + 1))
                : "v66", "v67", "v80", "v81", "v82", "v83", "v110", "v112", "v113", "s2");
        else
            asm volatile(
                "v_mov_b32 v80, 1.0\n\tv_mov_b32 v81, 1.0\n\tv_mov_b32 v82, 1.0\n\tv_mov_b32 v83, 1.0\n\t"
                "v_mov_b32 v112, %[b0]\n\tv_mov_b32 v113, %[b1]\n\tv_mov_b32 v110, %[k]\n\t"
                "s_nop 4\n\t"
                "v_mfma_f32_32x32x16_f16 %[c], %[a], v[80:83], %[c]\n\t"
                "v_mov_b32 v80, %[k]\n\t"
                "v_mul_f32 v66, v80, v112\n\t"
                "v_mul_f32 v80, v80, v113\n\t"
                "s_mulk_i32 s2, 0x4a00\n\t"
                "v_pk_fma_f32 %[acc01], %[x0a], v[66:67], %[acc01] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %[acc23], %[x0b], v[66:67], %[acc23] op_sel_hi:[1,0,1]\n\t"
                "v_fma_mixlo_f16 v66, %[k], v110, 0\n\t"
                "v_add_u32 v81, s2, v81\n\t"
                "v_fma_mixhi_f16 v66, %[b0], v110, 0\n\t"
                "v_pk_fma_f32 %[acc01], %[x1a], v[80:81], %[acc01] op_sel_hi:[1,0,1]\n\t"
                "v_fma_mixlo_f16 v67, %[k], v110, -v66 op_sel_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %[acc23], %[x1b], v[80:81], %[acc23] op_sel_hi:[1,0,1]"
                : [acc01] "+v"(*(f2 *)&acc), [acc23] "+v"(*((f2 *)&acc + 1)), [c] "+v"(c)
                : [a] "v"(a), [k] "v"(k), [b0] "v"(b), [b1] "v"(b1), [x0a] "v"(*(f2 *)&x0), [x0b] "v"(*((f2 *)&x0 + 1)),
                  [x1a] "v"(*(f2 *)&x1), [x1b] "v"(*((f2 *)&x1 + 1))
                : "v66", "v67", "v80", "v81", "v82", "v83", "v110", "v112", "v113", "s2");
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c[r];
    for (int e = 0; e < 4; ++e) out[(size_t)gid * 4 + e] = acc[e] + (s == 12345.f ? 1.f : 0.f);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(int iters, const float *__restrict__ in, float *__restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f2 acc = {0.f, 0.f};
    f16v c = {0};
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < iters; ++i) {
        const float k = in[(size_t)(3 * i + 0) * 64 + (threadIdx.x & 63)];
        const float b = in[(size_t)(3 * i + 1) * 64 + (threadIdx.x & 63)];
        const float xv = in[(size_t)(3 * i + 2) * 64 + (threadIdx.x & 63)];
        f2 x = {xv, xv + 1.f};
        if (VARIANT == 2) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b8, c, 0, 0, 0);
        if (VARIANT == 1)
            asm volatile("v_mul_f32 v100, %1, %2\n\tv_mov_b32 v102, v103\n\tv_pk_fma_f32 %0, %3, v[100:101], %0 op_sel_hi:[1,0,1]"
                         : "+v"(acc)
                         : "v"(k), "v"(b), "v"(x)
                         : "v100", "v101", "v102");
        else
            asm volatile("v_mul_f32 v100, %1, %2\n\tv_pk_fma_f32 %0, %3, v[100:101], %0 op_sel_hi:[1,0,1]"
                         : "+v"(acc)
                         : "v"(k), "v"(b), "v"(x)
                         : "v100", "v101");
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c[r];
    out[(size_t)gid * 2 + 0] = acc.x;
    out[(size_t)gid * 2 + 1] = acc.y + (s == 12345.f ? 1.f : 0.f);
}

int main() {
    const int iters = 4096, blocks = 4096, threads = 256;
    std::vector<float> in((size_t)3 * iters * 64);
    unsigned st = 12345u;
    for (auto &v : in) {
        st = st * 1664525u + 1013904223u;
        v = ((st >> 8) & 0xffff) / 65536.f + 0.25f;
    }
    float *din, *dout;
    hipMalloc((void **)&din, in.size() * 4);
    hipMalloc((void **)&dout, (size_t)blocks * threads * 2 * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    // reference: one wave's 64 lanes (every wave computes the same thing)
    std::vector<float> ref(128, 0.f);
    for (int l = 0; l < 64; ++l) {
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < iters; ++i) {
            const float k = in[(size_t)(3 * i + 0) * 64 + l], b = in[(size_t)(3 * i + 1) * 64 + l], x = in[(size_t)(3 * i + 2) * 64 + l];
            const float t = k * b;
            a0 = fmaf(x, t, a0);
            a1 = fmaf(x + 1.f, t, a1);
        }
        ref[2 * l] = a0;
        ref[2 * l + 1] = a1;
    }
    std::vector<float> out((size_t)blocks * threads * 2);
    for (int variant = 0; variant < 3; ++variant) {
        long bad = 0, launches = 20;
        for (int rep = 0; rep < launches; ++rep) {
            if (variant == 0) probe<0><<<blocks, threads>>>(iters, din, dout);
            else if (variant == 1) probe<1><<<blocks, threads>>>(iters, din, dout);
            else probe<2><<<blocks, threads>>>(iters, din, dout);
            hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
            for (size_t t = 0; t < (size_t)blocks * threads; ++t) {
                const int l = t & 63;
                if (out[2 * t] != ref[2 * l] || out[2 * t + 1] != ref[2 * l + 1]) ++bad;
            }
        }
        std::printf("{\"variant\": %d, \"launches\": %ld, \"lanes_per_launch\": %d, \"wrong_lane_results\": %ld}\n", variant, launches,
                    blocks * threads, bad);
    }
    // the kernel's own sequence
    float *dout4;
    hipMalloc((void **)&dout4, (size_t)blocks * threads * 4 * 4);
    std::vector<float> ref4(256, 0.f), out4((size_t)blocks * threads * 4);
    for (int l = 0; l < 64; ++l) {
        float a4[4] = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
            const float k = in[(size_t)(3 * i + 0) * 64 + l], b = in[(size_t)(3 * i + 1) * 64 + l], x = in[(size_t)(3 * i + 2) * 64 + l];
            const float t0 = k * b, t1 = k * (b + 0.5f);
            for (int e = 0; e < 4; ++e) a4[e] = fmaf(x + (float)e, t0, a4[e]);
            for (int e = 0; e < 4; ++e) a4[e] = fmaf(x + 4.f + (float)e, t1, a4[e]);
        }
        for (int e = 0; e < 4; ++e) ref4[4 * l + e] = a4[e];
    }
    for (int order : {1, 6}) {
        long bad = 0, launches = 20;
        for (int rep = 0; rep < launches; ++rep) {
            if (order == 1) probe_seq<1><<<blocks, threads>>>(iters, din, dout4);
            else probe_seq<6><<<blocks, threads>>>(iters, din, dout4);
            hipMemcpy(out4.data(), dout4, out4.size() * 4, hipMemcpyDeviceToHost);
            for (size_t t = 0; t < (size_t)blocks * threads; ++t)
                for (int e = 0; e < 4; ++e)
                    if (out4[4 * t + e] != ref4[4 * (t & 63) + e]) { ++bad; break; }
        }
        std::printf("{\"kernel_sequence_order\": %d, \"launches\": %ld, \"lanes_per_launch\": %d, \"wrong_lane_results\": %ld}\n", order,
                    launches, blocks * threads, bad);
    }
    return 0;
}
