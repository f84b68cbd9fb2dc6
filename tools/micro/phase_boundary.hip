// Micro-benchmark for the "one cooperative kernel per transformer block" question: what does a phase boundary cost on this
// part when the next phase reads what OTHER workgroups (other XCDs, other L2s) wrote in the previous one?
//   graph     : one kernel per phase, the chain captured in a hipGraph (what the engine does today)
//   barrier   : ONE persistent launch, a grid barrier between phases (agent-scope release before arriving, acquire after leaving:
//               the eight XCD L2s are not coherent with each other inside a kernel, so the release writes dirty lines back
//               (buffer_wbl2 sc1) and the acquire invalidates (buffer_inv sc1))
//   graph_nontemporal_stores : the graph form with `global_store ... nt` outputs (do they leave the L2 earlier than the end-of-kernel
//               write-back? no: the same to 1 %, slower at 64 MB per phase)
//   dataflow  : ONE persistent launch, no barrier: a workgroup waits only for the flag of the workgroup whose slice it reads
//               (same release / acquire pair per dependency)
// Every phase: workgroup i reads the SLICE bytes workgroup (i * 37 + 11) % G wrote in the previous phase, adds one, writes its
// own slice -- the minimum a dependent phase does; `work` extra passes over the slice stand in for a longer phase.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/phase_boundary.hip -o tools/micro/phase_boundary     Run on the GPU box.
// Output: CSV  mode, workgroups, threads, slice_kb, work, phases, us_per_phase, ok
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int src_of(int i, int G) { return (i * 37 + 11) % G; }

__device__ __forceinline__ void do_phase(const float4* in, float4* out, int i, int G, int n4, int work) {
    const float4* s = in + (size_t)src_of(i, G) * n4;
    float4* d = out + (size_t)i * n4;
    for (int k = threadIdx.x; k < n4; k += blockDim.x) {
        float4 v = s[k];
        for (int w = 0; w < work; ++w) {
            v.y = v.y * 1.0000001f + 1e-9f;
            asm volatile("" : "+v"(v.y));          // one dependent chain per element, the same code in every kernel
        }
        v.x += 1.f;
        d[k] = v;
    }
}

__global__ __launch_bounds__(512) void phase_kernel(const float4* in, float4* out, int n4, int work) {
    do_phase(in, out, blockIdx.x, gridDim.x, n4, work);
}

// the same phase with NON-TEMPORAL stores (global_store ... nt): the output does not sit dirty in the XCD's L2 until the end-of-kernel
// write-back, it streams out while the phase runs
typedef __attribute__((ext_vector_type(4))) float f4v;
__global__ __launch_bounds__(512) void phase_nt_kernel(const float4* in, float4* out, int n4, int work) {
    const int i = blockIdx.x, G = gridDim.x;
    const f4v* s = (const f4v*)in + (size_t)src_of(i, G) * n4;
    f4v* d = (f4v*)out + (size_t)i * n4;
    for (int k = threadIdx.x; k < n4; k += blockDim.x) {
        f4v v = s[k];
        for (int w = 0; w < work; ++w) {
            v.y = v.y * 1.0000001f + 1e-9f;
            asm volatile("" : "+v"(v.y));
        }
        v.x += 1.f;
        __builtin_nontemporal_store(v, d + k);
    }
}

__global__ __launch_bounds__(512) void barrier_kernel(float4* buf, unsigned* ctr, int n4, int work, int phases) {
    const int G = gridDim.x;
    const size_t half = (size_t)G * n4;
    for (int p = 0; p < phases; ++p) {
        do_phase(buf + (p & 1) * half, buf + ((p + 1) & 1) * half, blockIdx.x, G, n4, work);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(p + 1);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// the same grid barrier with ONE wave per workgroup issuing the release / acquire pair (the write-back and the invalidate act on the
// XCD's L2 and the CU's L1, not on a wave's own lines: once every wave's stores have left the CU -- vmcnt(0) ahead of the workgroup
// barrier -- one wave's fence covers the workgroup)
__global__ __launch_bounds__(512) void barrier1_kernel(float4* buf, unsigned* ctr, int n4, int work, int phases) {
    const int G = gridDim.x;
    const size_t half = (size_t)G * n4;
    for (int p = 0; p < phases; ++p) {
        do_phase(buf + (p & 1) * half, buf + ((p + 1) & 1) * half, blockIdx.x, G, n4, work);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(p + 1);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

// clock warm-up: ~100 ms of dependent FMAs on every CU before the timed launches
__global__ __launch_bounds__(512) void spin_kernel(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0000001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 1e-7f; b = b * 0.99999f + 1e-5f; }
    if (a == 123.456f) out[0] = a + b;
}

__global__ __launch_bounds__(512) void dataflow_kernel(float4* buf, unsigned* flags, int n4, int work, int phases) {
    const int G = gridDim.x;
    const size_t half = (size_t)G * n4;
    const int i = blockIdx.x;
    int r = 0;                                   // reader of my slice: r with src_of(r) == i (37 is invertible mod G)
    for (int j = 0; j < G; ++j) if (src_of(j, G) == i) r = j;
    for (int p = 0; p < phases; ++p) {
        if (p > 0) {
            // the slice this phase reads: written by src in phase p - 1 (flag value p); the slice this phase OVERWRITES was read in
            // phase p - 1 by the workgroups j with src_of(j) == i -- they are done with it once THEY finished phase p - 1, which
            // the reader of flag[j] >= p below cannot know in general; the benchmark keeps to the read dependency plus a
            // write-after-read wait on every reader (G is small: one reader per slice, the map is a permutation)
            if (threadIdx.x == 0) {
                const int s = src_of(i, G);
                while (__hip_atomic_load(flags + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p) __builtin_amdgcn_s_sleep(1);
            }
            if (threadIdx.x == 64) {
                // my phase-p write goes to buffer (p + 1) & 1, last read by r in phase p - 1 -> r must have finished phase p - 1
                while (__hip_atomic_load(flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        do_phase(buf + (p & 1) * half, buf + ((p + 1) & 1) * half, i, G, n4, work);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + i, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static bool check(const float4* dbuf, int G, int n4, int phases) {
    std::vector<float4> h((size_t)2 * G * n4);
    CK(hipMemcpy(h.data(), dbuf, h.size() * sizeof(float4), hipMemcpyDeviceToHost));
    const float4* fin = h.data() + (size_t)(phases & 1) * G * n4;
    for (size_t k = 0; k < (size_t)G * n4; ++k)
        if (fin[k].x != (float)phases) return false;
    return true;
}

int main(int argc, char** argv) {
    const int G = 256, T = 512;
    const int phases = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = 20;
    printf("mode,workgroups,threads,slice_kb,work,phases,us_per_phase,ok\n");
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned* ctr;
    CK(hipMalloc(&ctr, (G + 1) * sizeof(unsigned)));
    for (int slice_kb : {8, 64, 256}) {
        const int n4 = slice_kb * 1024 / 16;
        float4* buf;
        CK(hipMalloc(&buf, (size_t)2 * G * n4 * sizeof(float4)));
        for (int work : {0, 1000}) {
            // ---- graph of per-phase kernels
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < phases; ++p)
                hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(T), 0, st, buf + (size_t)(p & 1) * G * n4, buf + (size_t)((p + 1) & 1) * G * n4, n4, work);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float best[5] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
            bool ok[5] = {true, true, true, true, true};
            hipGraph_t g2;
            hipGraphExec_t ge2;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int p = 0; p < phases; ++p)
                hipLaunchKernelGGL(phase_nt_kernel, dim3(G), dim3(T), 0, st, buf + (size_t)(p & 1) * G * n4, buf + (size_t)((p + 1) & 1) * G * n4, n4, work);
            CK(hipStreamEndCapture(st, &g2));
            CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(512), 0, st, (float*)ctr, 4000000);
            for (int r = 0; r < reps; ++r) {
                float ms;
                CK(hipMemsetAsync(buf, 0, (size_t)2 * G * n4 * sizeof(float4), st));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[0]) best[0] = ms;
                if (r == 0) ok[0] = check(buf, G, n4, phases);

                CK(hipMemsetAsync(buf, 0, (size_t)2 * G * n4 * sizeof(float4), st));
                CK(hipMemsetAsync(ctr, 0, (G + 1) * sizeof(unsigned), st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(T), 0, st, buf, ctr + G, n4, work, phases);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[1]) best[1] = ms;
                if (r == 0) ok[1] = check(buf, G, n4, phases);

                CK(hipMemsetAsync(buf, 0, (size_t)2 * G * n4 * sizeof(float4), st));
                CK(hipMemsetAsync(ctr, 0, (G + 1) * sizeof(unsigned), st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(dataflow_kernel, dim3(G), dim3(T), 0, st, buf, ctr, n4, work, phases);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[2]) best[2] = ms;
                if (r == 0) ok[2] = check(buf, G, n4, phases);

                CK(hipMemsetAsync(buf, 0, (size_t)2 * G * n4 * sizeof(float4), st));
                CK(hipMemsetAsync(ctr, 0, (G + 1) * sizeof(unsigned), st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(barrier1_kernel, dim3(G), dim3(T), 0, st, buf, ctr + G, n4, work, phases);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[3]) best[3] = ms;
                if (r == 0) ok[3] = check(buf, G, n4, phases);

                CK(hipMemsetAsync(buf, 0, (size_t)2 * G * n4 * sizeof(float4), st));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge2, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[4]) best[4] = ms;
                if (r == 0) ok[4] = check(buf, G, n4, phases);
            }
            const char* names[5] = {"graph", "barrier", "dataflow", "barrier_one_fence", "graph_nontemporal_stores"};
            for (int m = 0; m < 5; ++m)
                printf("%s,%d,%d,%d,%d,%d,%.3f,%d\n", names[m], G, T, slice_kb, work, phases, best[m] * 1e3f / phases, ok[m] ? 1 : 0);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        CK(hipFree(buf));
    }
    return 0;
}
