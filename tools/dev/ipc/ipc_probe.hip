// dev probe: what works between two PROCESSES sharing one GPU (or two GPUs) on this stack?
//   hipIpcGetMemHandle / hipIpcOpenMemHandle on ordinary and fine-grained device memory, hipMemcpyAsync into the peer's buffer,
//   a flag word written behind the copy (hipStreamWriteValue64 | a one-lane kernel) and awaited on the reader's stream
//   (hipStreamWaitValue64 | a spinning one-lane kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] HIP error %s (%d) at line %d: %s\n", getpid(), hipGetErrorString(e_), (int)e_, __LINE__, #x); fflush(stdout); exit(1); } } while (0)
#define TRY(x) ({ hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); (void)hipGetLastError(); } e_; })

__global__ void flag_store(unsigned long long *f, unsigned long long v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void flag_wait(unsigned long long *f, unsigned long long v, unsigned long long *spins)
{
    unsigned long long n = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) { __builtin_amdgcn_s_sleep(16); ++n; }
    if (spins) *spins = n;
}
__global__ void fill(double *p, size_t n, double v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v + (double)i; }
__global__ void check(const double *p, size_t n, double v, int *bad) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n && p[i] != v + (double)i) atomicAdd(bad, 1); }

struct Blob { hipIpcMemHandle_t data, flag; int pid; };

int main(int argc, char **argv)
{
    const size_t n = 1 << 21;      // 16 MB message
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    const pid_t child = fork();     // before any HIP call
    const bool reader = child == 0;
    int ndev = 0; CK(hipGetDeviceCount(&ndev));
    const int dev = (ndev > 1 && reader) ? 1 : 0;
    CK(hipSetDevice(dev));
    int canwait = 0; (void)hipDeviceGetAttribute(&canwait, hipDeviceAttributeCanUseStreamWaitValue, dev);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (reader) {
        double *buf; unsigned long long *flag, *spins; int *bad;
        CK(hipMalloc(&buf, n * 8)); CK(hipMemset(buf, 0, n * 8));
        hipError_t fg = TRY(hipExtMallocWithFlags((void **)&flag, 4096, hipDeviceMallocFinegrained));
        if (fg != hipSuccess) CK(hipMalloc(&flag, 4096));
        CK(hipMemset(flag, 0, 4096)); CK(hipMalloc(&spins, 8)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        CK(hipDeviceSynchronize());
        Blob b; b.pid = getpid();
        CK(hipIpcGetMemHandle(&b.data, buf)); CK(hipIpcGetMemHandle(&b.flag, flag));
        printf("[reader %d] dev %d of %d, can_use_stream_wait_value=%d, fine-grained flag %s, handles exported\n", getpid(), dev, ndev, canwait, fg == hipSuccess ? "yes" : "NO (coarse)"); fflush(stdout);
        if (write(c2p[1], &b, sizeof b) != sizeof b) return 1;
        for (unsigned long long round = 1; round <= 4; ++round) {
            const bool use_waitvalue = canwait && (round & 1) == 0;
            if (use_waitvalue) { if (TRY(hipStreamWaitValue64(st, flag, round, hipStreamWaitValueGte, ~0ull)) != hipSuccess) flag_wait<<<1, 1, 0, st>>>(flag, round, spins); }
            else flag_wait<<<1, 1, 0, st>>>(flag, round, spins);
            check<<<(n + 255) / 256, 256, 0, st>>>(buf, n, (double)round, bad);
            char go = 1; if (write(c2p[1], &go, 1) != 1) return 1;       // the wait is enqueued: now let the writer send
            CK(hipStreamSynchronize(st));
            int hb = -1; unsigned long long hs = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, spins, 8, hipMemcpyDeviceToHost));
            printf("[reader] round %llu (%s): mismatches %d, spins %llu\n", round, use_waitvalue ? "hipStreamWaitValue64" : "spin kernel", hb, hs); fflush(stdout);
            CK(hipMemset(bad, 0, 4));
        }
        return 0;
    }
    Blob b;
    if (read(c2p[0], &b, sizeof b) != sizeof b) return 1;
    double *peer = nullptr, *src; unsigned long long *pflag = nullptr;
    CK(hipIpcOpenMemHandle((void **)&peer, b.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void **)&pflag, b.flag, hipIpcMemLazyEnablePeerAccess));
    CK(hipMalloc(&src, n * 8));
    printf("[writer %d] dev %d, peer buffer %p flag %p opened\n", getpid(), dev, (void *)peer, (void *)pflag); fflush(stdout);
    for (unsigned long long round = 1; round <= 4; ++round) {
        char go; if (read(c2p[0], &go, 1) != 1) return 1;
        usleep(200000);          // the reader's wait is really waiting
        fill<<<(n + 255) / 256, 256, 0, st>>>(src, n, (double)round);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        CK(hipMemcpyAsync(peer, src, n * 8, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e1, st));
        const bool use_writevalue = round >= 3;
        if (!use_writevalue || TRY(hipStreamWriteValue64(st, pflag, round, 0)) != hipSuccess) flag_store<<<1, 1, 0, st>>>(pflag, round);
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("[writer] round %llu: copy %.3f ms = %.1f GB/s, flag by %s\n", round, ms, n * 8 / ms / 1e6, use_writevalue ? "hipStreamWriteValue64" : "kernel"); fflush(stdout);
    }
    int status = 0; waitpid(child, &status, 0);
    printf("[writer] reader exit status %d\n", WEXITSTATUS(status));
    return 0;
}
