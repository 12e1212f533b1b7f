// rk3d_transport.h -- the slab exchange's transports inside the library (include/lbmpm.h, "Transport of the slab exchange"):
// IPC landing areas filled by copy-engine transfers and stream value operations, or ncclSend / ncclRecv of a librccl opened at run
// time.  Included by rk3d.hip (host code only; the two one-lane kernels are the fallback of devices without stream value operations).
#include <dlfcn.h>
#include <unistd.h>
#include <time.h>
#include <stdio.h>

namespace slabtx {

using lbmpm::set_error;

__global__ void flag_store(unsigned long long *f, unsigned long long v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void flag_wait(unsigned long long *f, unsigned long long v)
{
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) __builtin_amdgcn_s_sleep(32);
}

// what a rank tells its neighbours (LBMPM_IPC_BLOB_BYTES)
struct IpcBlob {
    uint32_t magic, version;
    int32_t pid, device;
    uint64_t slot_bytes;                  // bytes between the four slots of the landing area: [from below | from above][parity]
    uint64_t bytes_from_below, bytes_from_above;      // message sizes this rank expects (0: no neighbour there)
    uint64_t land_ptr, flags_ptr;         // addresses in the owner's process (used by slabs of the same process)
    hipIpcMemHandle_t land, flags;
    uint64_t nonce;                       // drawn once per process: "same pid" alone does not mean "same process" (ranks in separate
                                          // containers or pid namespaces of one node commonly share a pid, e.g. 1)
};
static_assert(sizeof(IpcBlob) <= LBMPM_IPC_BLOB_BYTES, "blob size is part of the ABI");
constexpr uint32_t BLOB_MAGIC = 0x4c424d50u;     // "LBMP"
constexpr uint32_t BLOB_VERSION = 2;

// one random 64-bit value per process (from /dev/urandom; pid, clock and an address as the fallback)
inline uint64_t process_nonce()
{
    static uint64_t n = 0;
    if (n) return n;
    uint64_t v = 0;
    if (FILE *f = fopen("/dev/urandom", "rb")) { if (fread(&v, sizeof v, 1, f) != 1) v = 0; fclose(f); }
    if (!v) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        v = ((uint64_t)getpid() << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)ts.tv_sec << 20) ^ reinterpret_cast<uint64_t>(&n);
    }
    n = v | 1ull;
    return n;
}

// the part of librccl this file calls (types as in rccl.h: opaque communicator, 128-byte id, int enums)
struct Rccl {
    void *dl = nullptr;
    typedef struct { char internal[LBMPM_RCCL_ID_BYTES]; } UniqueId;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;          // optional (present in every RCCL since 2.4): tears a communicator down without its peers
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static constexpr int kFloat64 = 8;        // ncclFloat64 / ncclDouble
    int open(const char *path)
    {
        const char *names[] = {path, "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (dl) break;
        }
        if (!dl) { set_error("RCCL transport: cannot open librccl (%s)", dlerror()); return LBMPM_ERR_UNSUPPORTED; }
        bool ok = true;
        auto sym = [&](const char *n) { void *s = dlsym(dl, n); if (!s) ok = false; return s; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(dl, "ncclCommAbort"));
        if (!ok) { set_error("RCCL transport: librccl lacks ncclSend / ncclRecv / ncclCommInitRank"); close(); return LBMPM_ERR_UNSUPPORTED; }
        return LBMPM_OK;
    }
    void close() { if (dl) dlclose(dl); dl = nullptr; }
};

struct Transport {
    int kind = LBMPM_TRANSPORT_NONE;
    int device = 0;
    bool has_below = false, has_above = false;
    size_t bytes_up = 0, bytes_dn = 0;        // message to the rank above (this rank's top plane) / below (its bottom plane)
    size_t bytes_from_below = 0, bytes_from_above = 0;     // messages from there (their face planes = this rank's halo planes)
    unsigned long long seq = 0;               // messages exchanged so far (all ranks count alike)
    // ---- IPC
    char *land = nullptr;                     // own landing area, 4 slots
    size_t slot = 0;
    bool land_fine = false;                   // the landing area is fine-grained memory
    unsigned long long *flags = nullptr;      // own, fine-grained: [face 0 from below | 1 from above][parity]
    char *peer_land[2] = {nullptr, nullptr};  // [0] landing area of the rank below (we fill its "from above" slots), [1] of the rank above
    unsigned long long *peer_flags[2] = {nullptr, nullptr};
    size_t peer_slot[2] = {0, 0};             // the neighbours' slot sizes (their halo planes differ from ours)
    bool mapped[2] = {false, false};          // peer_* came from hipIpcOpenMemHandle (to be closed)
    bool value_ops = false;
    bool connected = false;
    bool dead = false;                        // the steady-state watchdog gave up on a neighbour (lbmpm_rk3d_sync_deadline)
    // ---- RCCL
    Rccl rccl;
    void *comm = nullptr;
    int rank = 0, nranks = 1;
    int peer_up = -1, peer_dn = -1;           // RCCL ranks of the neighbours (rank + 1, rank - 1; the self-test talks to itself)

    char *slot_ptr(char *base, int face, unsigned par) const { return base + ((size_t)face * 2 + par) * slot; }
    char *peer_slot_ptr(int side, int face, unsigned par) const { return peer_land[side] + ((size_t)face * 2 + par) * peer_slot[side]; }

    int set_shape(int dev, bool below, bool above, size_t up, size_t dn, size_t from_below, size_t from_above)
    {
        if (kind != LBMPM_TRANSPORT_NONE) { set_error("a transport is connected already: lbmpm_rk3d_transport_disconnect first"); return LBMPM_ERR_STATE; }
        device = dev; has_below = below; has_above = above; bytes_up = up; bytes_dn = dn; bytes_from_below = from_below; bytes_from_above = from_above;
        const size_t m = from_below > from_above ? from_below : from_above;
        slot = (m + 4095) / 4096 * 4096;
        LBMPM_HIP_TRY(hipSetDevice(dev));
        if (land) { (void)hipFree(land); land = nullptr; }          // (a connect that failed half way and is tried again)
        // The landing area is FINE-GRAINED memory where the device offers it: with the IPC transport a neighbour GPU's copy engine writes it
        // over xGMI, past this GPU's L2, and ordinary (coarse-grained) memory is coherent at kernel boundaries only -- a slot is reused
        // every second step, and a line of it left in L2 by the previous unpack would be served stale.  (The probe at set-up would catch
        // that and the selection would fall back to RCCL; this keeps the copy-engine path.  LBMPM_IPC_LAND=coarse: ordinary memory.)
#ifdef LBMPM_DEV
        const char *lk = getenv("LBMPM_IPC_LAND");
#else
        const char *lk = nullptr;
#endif
        land_fine = !(lk && !strcmp(lk, "coarse")) &&
                    hipExtMallocWithFlags(reinterpret_cast<void **>(&land), 4 * slot, hipDeviceMallocFinegrained) == hipSuccess;
        if (!land_fine) {
            (void)hipGetLastError();
            LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&land), 4 * slot));
        }
        return LBMPM_OK;
    }

    int ipc_alloc(IpcBlob *blob)
    {
        const int dev = device;
        const bool below = has_below, above = has_above;
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&flags), 4096, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            LBMPM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&flags), 4096));
        }
        LBMPM_HIP_TRY(hipMemset(flags, 0, 4096));
        LBMPM_HIP_TRY(hipDeviceSynchronize());
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev);
        value_ops = can != 0 && !getenv("LBMPM_IPC_FLAG_KERNELS");      // (the variable forces the one-lane kernels: test coverage of the fallback)
        memset(blob, 0, sizeof *blob);
        blob->magic = BLOB_MAGIC; blob->version = BLOB_VERSION; blob->pid = (int32_t)getpid(); blob->device = dev;
        blob->nonce = process_nonce();
        blob->slot_bytes = slot; blob->bytes_from_below = below ? bytes_from_below : 0; blob->bytes_from_above = above ? bytes_from_above : 0;
        blob->land_ptr = reinterpret_cast<uint64_t>(land); blob->flags_ptr = reinterpret_cast<uint64_t>(flags);
        LBMPM_HIP_TRY(hipIpcGetMemHandle(&blob->land, land));
        LBMPM_HIP_TRY(hipIpcGetMemHandle(&blob->flags, flags));
        kind = LBMPM_TRANSPORT_IPC; seq = 0; connected = false;
        return LBMPM_OK;
    }

    int ipc_open(int side, const IpcBlob *b, size_t my_bytes)
    {
        if (b->magic != BLOB_MAGIC || b->version != BLOB_VERSION) { set_error("lbmpm_rk3d_ipc_connect: not a blob of lbmpm_rk3d_ipc_init"); return LBMPM_ERR_INVALID; }
        const uint64_t theirs = side == 0 ? b->bytes_from_above : b->bytes_from_below;     // the rank below receives "from above"
        if (theirs != my_bytes) {
            set_error("lbmpm_rk3d_ipc_connect: the rank %s expects %llu bytes per message, this rank sends %llu (different cuts or lattices)",
                      side == 0 ? "below" : "above", (unsigned long long)theirs, (unsigned long long)my_bytes);
            return LBMPM_ERR_INVALID;
        }
        if (b->slot_bytes < my_bytes) { set_error("lbmpm_rk3d_ipc_connect: the neighbour's slots are smaller than the message"); return LBMPM_ERR_INVALID; }
        peer_slot[side] = (size_t)b->slot_bytes;
        if (b->pid == (int32_t)getpid() && b->nonce == process_nonce()) {      // a slab of this very process: plain pointers (peer access if it lives on another GPU)
            if (b->device != device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess(%d): %s", b->device, hipGetErrorString(e)); return LBMPM_ERR_HIP; }
                (void)hipGetLastError();
            }
            peer_land[side] = reinterpret_cast<char *>(b->land_ptr);
            peer_flags[side] = reinterpret_cast<unsigned long long *>(b->flags_ptr);
            mapped[side] = false;
            return LBMPM_OK;
        }
        void *pl = nullptr, *pf = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&pl, b->land, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf, b->flags, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            if (pl) (void)hipIpcCloseMemHandle(pl);
            set_error("hipIpcOpenMemHandle (rank %s, pid %d, device %d): %s", side == 0 ? "below" : "above", b->pid, b->device, hipGetErrorString(e));
            (void)hipGetLastError();
            return LBMPM_ERR_HIP;
        }
        peer_land[side] = static_cast<char *>(pl); peer_flags[side] = static_cast<unsigned long long *>(pf); mapped[side] = true;
        return LBMPM_OK;
    }

    // One message each way, enqueued on `st`: send_up -> the rank above, send_dn -> the rank below; *from_below / *from_above = where
    // this rank's incoming messages will have landed when the stream gets past the waits enqueued here.
    int exchange(hipStream_t st, const double *send_up, const double *send_dn, const double **from_below, const double **from_above)
    {
        if (!connected) { set_error("the slab's transport is not connected"); return LBMPM_ERR_STATE; }
        if (dead) { set_error("the slab's transport was given up by the watchdog (a neighbour did not answer): disconnect and set up the run again"); return LBMPM_ERR_TIMEOUT; }
        seq += 1;
        const unsigned par = (unsigned)(seq & 1ull);
        if (kind == LBMPM_TRANSPORT_IPC) {
            *from_below = reinterpret_cast<const double *>(slot_ptr(land, 0, par));
            *from_above = reinterpret_cast<const double *>(slot_ptr(land, 1, par));
            if (has_above) LBMPM_HIP_TRY(hipMemcpyAsync(peer_slot_ptr(1, 0, par), send_up, bytes_up, hipMemcpyDeviceToDevice, st));
            if (has_below) LBMPM_HIP_TRY(hipMemcpyAsync(peer_slot_ptr(0, 1, par), send_dn, bytes_dn, hipMemcpyDeviceToDevice, st));
            auto post = [&](unsigned long long *f) -> hipError_t {
                if (value_ops) return hipStreamWriteValue64(st, f, seq, 0);
                flag_store<<<1, 1, 0, st>>>(f, seq);
                return hipGetLastError();
            };
            auto await = [&](unsigned long long *f) -> hipError_t {
                if (value_ops) return hipStreamWaitValue64(st, f, seq, hipStreamWaitValueGte, ~0ull);
                flag_wait<<<1, 1, 0, st>>>(f, seq);
                return hipGetLastError();
            };
            if (has_above) LBMPM_HIP_TRY(post(peer_flags[1] + 0 * 2 + par));
            if (has_below) LBMPM_HIP_TRY(post(peer_flags[0] + 1 * 2 + par));
            if (has_below) LBMPM_HIP_TRY(await(flags + 0 * 2 + par));
            if (has_above) LBMPM_HIP_TRY(await(flags + 1 * 2 + par));
            return LBMPM_OK;
        }
        if (kind == LBMPM_TRANSPORT_RCCL) {
            *from_below = reinterpret_cast<const double *>(slot_ptr(land, 0, 0));
            *from_above = reinterpret_cast<const double *>(slot_ptr(land, 1, 0));
            int rc = rccl.GroupStart();
            if (rc == 0 && has_above) rc = rccl.Send(send_up, bytes_up / 8, Rccl::kFloat64, peer_up, comm, st);
            if (rc == 0 && has_above) rc = rccl.Recv(slot_ptr(land, 1, 0), bytes_from_above / 8, Rccl::kFloat64, peer_up, comm, st);
            if (rc == 0 && has_below) rc = rccl.Send(send_dn, bytes_dn / 8, Rccl::kFloat64, peer_dn, comm, st);
            if (rc == 0 && has_below) rc = rccl.Recv(slot_ptr(land, 0, 0), bytes_from_below / 8, Rccl::kFloat64, peer_dn, comm, st);
            const int rc2 = rccl.GroupEnd();
            if (rc == 0) rc = rc2;
            if (rc != 0) { set_error("RCCL transport: %s", rccl.GetErrorString ? rccl.GetErrorString(rc) : "ncclSend / ncclRecv failed"); return LBMPM_ERR_HIP; }
            return LBMPM_OK;
        }
        set_error("no transport");
        return LBMPM_ERR_STATE;
    }

    void disconnect()
    {
        for (int s = 0; s < 2; ++s) {
            if (mapped[s]) { (void)hipIpcCloseMemHandle(peer_land[s]); (void)hipIpcCloseMemHandle(peer_flags[s]); }
            peer_land[s] = nullptr; peer_flags[s] = nullptr; mapped[s] = false;
        }
        if (comm) {         // dead: a peer is known not to answer (the watchdog fired) -- ncclCommDestroy would wait for it
            if (dead && rccl.CommAbort) (void)rccl.CommAbort(comm); else if (!dead) (void)rccl.CommDestroy(comm);
            comm = nullptr;
        }
        rccl.close();
        if (land) (void)hipFree(land);
        if (flags) (void)hipFree(flags);
        land = nullptr; flags = nullptr;
        kind = LBMPM_TRANSPORT_NONE; connected = false; seq = 0; dead = false;
        (void)hipGetLastError();
    }
};

}  // namespace slabtx
