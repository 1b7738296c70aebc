// hostcopy.hip — what the C ABI's host hand-over costs on this box: pageable vs pinned H2D / D2H copies,
// hipMalloc / hipFree of large buffers, a threaded pageable->pinned staging pipeline.
// Build: hipcc -O2 --offload-arch=gfx950 -o hostcopy hostcopy.hip -lpthread ; run: ./hostcopy [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t n = (size_t)((argc > 1 ? std::atof(argv[1]) : 1.0) * (1ull << 30));
    OK(hipSetDevice(0));
    std::vector<uint8_t> pageable(n, 1);
    uint8_t *pinned = nullptr, *dev = nullptr;
    double t = now();
    OK(hipHostMalloc((void **)&pinned, n, hipHostMallocDefault));
    std::printf("hipHostMalloc %.2f GiB: %.1f ms\n", n / 1073741824.0, (now() - t) * 1e3);
    std::memset(pinned, 2, n);
    t = now(); OK(hipMalloc((void **)&dev, n)); std::printf("hipMalloc %.2f GiB: %.2f ms\n", n / 1073741824.0, (now() - t) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); OK(hipMemcpy(dev, pageable.data(), n, hipMemcpyHostToDevice)); double a = now() - t;
        t = now(); OK(hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice)); double b = now() - t;
        t = now(); OK(hipMemcpy(pageable.data(), dev, n, hipMemcpyDeviceToHost)); double c = now() - t;
        t = now(); OK(hipMemcpy(pinned, dev, n, hipMemcpyDeviceToHost)); double d = now() - t;
        std::printf("rep %d: H2D pageable %.1f GB/s, H2D pinned %.1f GB/s, D2H pageable %.1f GB/s, D2H pinned %.1f GB/s\n", rep,
                    n / a / 1e9, n / b / 1e9, n / c / 1e9, n / d / 1e9);
    }
    // threaded staging: T threads memcpy 8 MiB chunks pageable -> pinned ring, each chunk sent with hipMemcpyAsync
    for (int T : {1, 2, 4, 8}) {
        const size_t chunk = 8u << 20;
        const size_t nchunks = (n + chunk - 1) / chunk;
        hipStream_t st; OK(hipStreamCreate(&st));
        t = now();
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back([&, k] {
            OK(hipSetDevice(0));
            hipStream_t s; OK(hipStreamCreate(&s));
            for (size_t c = k; c < nchunks; c += T) {
                const size_t o = c * chunk, len = std::min(chunk, n - o);
                std::memcpy(pinned + o, pageable.data() + o, len);       // (a real ring would reuse a few chunks)
                OK(hipMemcpyAsync(dev + o, pinned + o, len, hipMemcpyHostToDevice, s));
            }
            OK(hipStreamSynchronize(s)); OK(hipStreamDestroy(s));
        });
        for (auto &x : th) x.join();
        std::printf("staged H2D, %d threads: %.1f GB/s\n", T, n / (now() - t) / 1e9);
        OK(hipStreamDestroy(st));
    }
    for (int T : {1, 4, 8}) {
        const size_t chunk = 8u << 20;
        const size_t nchunks = (n + chunk - 1) / chunk;
        t = now();
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back([&, k] {
            OK(hipSetDevice(0));
            hipStream_t s; OK(hipStreamCreate(&s));
            for (size_t c = k; c < nchunks; c += T) {
                const size_t o = c * chunk, len = std::min(chunk, n - o);
                OK(hipMemcpyAsync(pinned + o, dev + o, len, hipMemcpyDeviceToHost, s));
                OK(hipStreamSynchronize(s));
                std::memcpy(pageable.data() + o, pinned + o, len);
            }
            OK(hipStreamDestroy(s));
        });
        for (auto &x : th) x.join();
        std::printf("staged D2H, %d threads: %.1f GB/s\n", T, n / (now() - t) / 1e9);
    }
    OK(hipFree(dev));
    for (size_t g : {1ull, 4ull, 8ull}) {
        void *p = nullptr;
        t = now(); OK(hipMalloc(&p, g << 30)); double a = now() - t;
        t = now(); OK(hipMemset(p, 0, g << 30)); OK(hipDeviceSynchronize()); double m = now() - t;
        t = now(); OK(hipFree(p)); double f = now() - t;
        std::printf("hipMalloc %zu GiB: %.2f ms, first memset %.2f ms, hipFree %.2f ms\n", g, a * 1e3, m * 1e3, f * 1e3);
    }
    return 0;
}
