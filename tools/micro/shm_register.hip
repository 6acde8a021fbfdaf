// micro-benchmark (development tool): can a POSIX shared-memory mapping be page-locked with hipHostRegister, and what H2D / D2H rate does it give?
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 800u << 20;
    const char* nm = "/foamyade_shm_probe";
    shm_unlink(nm);
    int fd = shm_open(nm, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { perror("shm"); return 1; }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { perror("mmap"); return 1; }
    memset(p, 1, bytes);
    void* d; hipMalloc(&d, bytes);
    hipStream_t s; hipStreamCreate(&s);
    for (int reg = 0; reg < 2; ++reg) {
        if (reg) {
            const double t0 = now();
            hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
            printf("hipHostRegister(shm, 800 MiB): %s in %.1f ms\n", hipGetErrorString(e), now() - t0);
            if (e != hipSuccess) break;
        }
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
            const double h2d = now() - t0;
            t0 = now();
            hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
            const double d2h = now() - t0;
            printf("%s: H2D %.1f ms = %.1f GB/s, D2H %.1f ms = %.1f GB/s\n", reg ? "registered" : "pageable  ", h2d, bytes / h2d * 1e-6, d2h, bytes / d2h * 1e-6);
        }
    }
    hipHostUnregister(p);
    munmap(p, bytes); close(fd); shm_unlink(nm);
    return 0;
}
