// Round trip host -> resident lane -> host, by where the doorbell and the payload live (the single-instance host call).
//   host: doorbell and 37 doubles of payload in mapped HOST memory (the lane polls and reads over the bus)
//   vram: doorbell and payload in fine-grained DEVICE memory that the host writes through the PCIe aperture (the lane polls locally)
// The lane answers with 118 doubles + a ticket in mapped host memory in both cases.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/resident_pingpong tools/resident_pingpong.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <csignal>
#include <immintrin.h>
#include <csetjmp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            std::exit(2);                                                              \
        }                                                                              \
    } while (0)

template <bool PRELOAD, bool COOPERATIVE, bool COMPUTE>
__global__ __launch_bounds__(64) void Serve(const double* in, double* out, unsigned long long* ring, unsigned long long* ack, unsigned long long served, unsigned long long life) {
    __shared__ double stage[128];
    const unsigned long long born = wall_clock64();
    for (;;) {
        unsigned long long asked = 0;
        if (threadIdx.x == 0) {
            for (;;) {
                asked = __hip_atomic_load(ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (asked != served || wall_clock64() - born > life) break;
            }
        }
        asked = __shfl(asked, 0);
        if (asked == served) return;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const double* p = in;
        asm volatile("" : "+v"(p) : : "memory");
        double x[37];
        if (COOPERATIVE) {
            if (threadIdx.x < 37) stage[threadIdx.x] = p[threadIdx.x];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 37; ++i) x[i] = stage[i];
        } else if (PRELOAD) {
#pragma unroll
            for (int i = 0; i < 37; ++i) x[i] = p[i];
        }
        if (threadIdx.x == 0 || COOPERATIVE) {
            double acc = 0.0;
            // a dependent chain about as long as the quadrotor node Jacobian's, reading its inputs where it needs them
#pragma unroll
            for (int i = 0; i < 37; ++i) {
                const double v = (PRELOAD || COOPERATIVE) ? x[i] : p[i];
                acc = __builtin_fma(acc, 0.999, v);
                if (COMPUTE) {
                    acc = __builtin_fma(acc, acc * 1e-3, v);
                    acc = __builtin_fma(acc, 1.0001, sin(v));
                }
            }
            if (COOPERATIVE) {
                __syncthreads();
                for (int k = threadIdx.x; k < 118; k += 64) out[k] = acc + k;
            } else {
#pragma unroll
                for (int k = 0; k < 118; k += 2) *reinterpret_cast<double2*>(out + k) = double2{acc + k, acc + k + 1};
            }
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
        if (COOPERATIVE) __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ack, asked, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        served = asked;
    }
}

template <bool PRELOAD, bool COOPERATIVE, bool COMPUTE = true>
double Run(const char* label, double* inHost, double* inDevice, unsigned long long* ringHost, unsigned long long* ringDevice, double* outHost, double* outDevice, unsigned long long* ackHost,
           unsigned long long* ackDevice, unsigned long long& ticket) {
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int calls = 20000;
    hipLaunchKernelGGL((Serve<PRELOAD, COOPERATIVE, COMPUTE>), dim3(1), dim3(64), 0, s, inDevice, outDevice, ringDevice, ackDevice, ticket, 100000000ull / 2);  // 0.5 s of life
    double payload[37];
    for (int i = 0; i < 37; ++i) payload[i] = 0.01 * i;
    double sum = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < calls; ++c) {
        payload[0] = c * 1e-6;
        std::memcpy(inHost, payload, sizeof payload);
        ++ticket;
        __atomic_store_n(ringHost, ticket, __ATOMIC_RELEASE);
        _mm_sfence();  // (a doorbell behind the PCIe aperture is write-combined: pushed out now, not when the buffer fills)
        volatile unsigned long long* a = ackHost;
        unsigned long long spins = 0;
        while (*a != ticket)
            if (++spins > 2000000000ull) {
                std::printf("%s: no answer\n", label);
                std::fflush(stdout);
                _exit(0);
            }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        sum += outHost[117];
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
    CHECK(hipStreamSynchronize(s));
    CHECK(hipStreamDestroy(s));
    std::printf("%-44s %6.2f us per call (checksum %.6g)\n", label, us, sum);
    std::fflush(stdout);
    return us;
}

int main() {
    double *inHost, *outHost, *inHostDevice, *outDevice;
    unsigned long long *mailHost, *mailDevice;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&inHost), 4096, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&inHostDevice), inHost, 0));
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&outHost), 4096, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&outDevice), outHost, 0));
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&mailHost), 4096, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&mailDevice), mailHost, 0));
    std::memset(mailHost, 0, 4096);
    unsigned long long ticket = 0;
    Run<false, false>("host mailbox, inputs read where used", inHost, inHostDevice, mailHost + 8, mailDevice + 8, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, false>("host mailbox, inputs requested up front", inHost, inHostDevice, mailHost + 8, mailDevice + 8, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, false, false>("host mailbox, up front, almost no arithmetic", inHost, inHostDevice, mailHost + 8, mailDevice + 8, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, true, false>("host mailbox, 64 lanes, almost no arithmetic", inHost, inHostDevice, mailHost + 8, mailDevice + 8, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, true>("host mailbox, 64 lanes fetch and store", inHost, inHostDevice, mailHost + 8, mailDevice + 8, outHost, outDevice, mailHost, mailDevice, ticket);
    // doorbell + payload in device memory the host can write
    void* vram = nullptr;
    if (hipExtMallocWithFlags(&vram, 8192, hipDeviceMallocFinegrained) != hipSuccess) {
        std::printf("fine-grained device memory: not available (%s)\n", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    CHECK(hipMemset(vram, 0, 8192));
    CHECK(hipDeviceSynchronize());
    std::fflush(stdout);
    static sigjmp_buf back;  // (a host store to memory that is not mapped for the CPU raises a signal: reported, not fatal)
    struct sigaction action {}, oldSegv {}, oldBus {};
    action.sa_handler = [](int) { siglongjmp(back, 1); };
    sigaction(SIGSEGV, &action, &oldSegv);
    sigaction(SIGBUS, &action, &oldBus);
    bool writable = false;
    if (sigsetjmp(back, 1) == 0) {
        volatile unsigned long long* probe = static_cast<volatile unsigned long long*>(vram);
        probe[512] = 7;
        writable = probe[512] == 7;
    }
    sigaction(SIGSEGV, &oldSegv, nullptr);
    sigaction(SIGBUS, &oldBus, nullptr);
    if (!writable) {
        std::printf("fine-grained device memory is not writable by the host here\n");
        return 0;
    }
    double* inVram = static_cast<double*>(vram);
    unsigned long long* ringVram = reinterpret_cast<unsigned long long*>(static_cast<char*>(vram) + 4096);
    ticket = 0;
    mailHost[0] = 0;
    Run<false, false>("VRAM doorbell + inputs, read where used", inVram, inVram, ringVram, ringVram, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, false>("VRAM doorbell + inputs, requested up front", inVram, inVram, ringVram, ringVram, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, true, false>("VRAM doorbell + inputs, 64 lanes, no arithmetic", inVram, inVram, ringVram, ringVram, outHost, outDevice, mailHost, mailDevice, ticket);
    Run<true, true>("VRAM doorbell + inputs, 64 lanes", inVram, inVram, ringVram, ringVram, outHost, outDevice, mailHost, mailDevice, ticket);
    return 0;
}
