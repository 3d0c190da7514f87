// Does hipStreamWaitValue32 work here, and does a kernel's system-scope store release a stream that waits on it while the kernel is still running?
// hipcc --offload-arch=gfx950 -O2 tools/probe_stream_wait.hip -o /tmp/probe_stream_wait && /tmp/probe_stream_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(uint32_t* signal, uint32_t value, unsigned long long* stamps, int spinCycles) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[0] = wall_clock64();
        __threadfence();
        __hip_atomic_store(signal, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spinCycles) {}   // keep running long after the signal
        stamps[1] = wall_clock64();
    }
}
__global__ void consumer(unsigned long long* stamps) { if (threadIdx.x == 0) stamps[2] = wall_clock64(); }
int main() {
    uint32_t* signal = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&signal, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    CHECK(hipMemset(signal, 0, 8));
    unsigned long long* stamps;
    CHECK(hipHostMalloc((void**)&stamps, 64, hipHostMallocDefault));
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int round = 1; round <= 3; round++) {
        stamps[0] = stamps[1] = stamps[2] = 0;
        e = hipStreamWaitValue32(b, signal, (uint32_t)round, hipStreamWaitValueGte, 0xffffffffu);
        printf("round %d hipStreamWaitValue32: %s\n", round, hipGetErrorString(e));
        if (e != hipSuccess) return 1;
        consumer<<<1, 64, 0, b>>>(stamps);
        producer<<<1, 64, 0, a>>>(signal, (uint32_t)round, stamps, 20000000); // ~200 ms at 100 MHz wall clock
        CHECK(hipStreamSynchronize(b));
        const bool early = stamps[1] == 0; // the consumer finished while the producer was still spinning
        CHECK(hipStreamSynchronize(a));
        printf("round %d: consumer ran %s the producer ended; signal -> consumer %.1f us (wall clock 100 MHz)\n", round, early ? "BEFORE" : "after",
               (double)(stamps[2] - stamps[0]) / 100.0);
    }
    return 0;
}
