// fsnap_p2p.cpp -- second transport of the multi-GPU exchange step: a one-shot all-reduce over peer-to-peer mapped
// device memory (hipIpc handles; xGMI between GPUs of one node) with the small host-side collectives through POSIX
// shared memory.  Compiled as HIP.
//
// Replaces, like the RCCL transport in fsnap_comm.cpp:
//   comm.Allreduce(c), comm.Allreduce(d)     examples/library/transpose_trick/example.py:245-246
//   comm.bcast / comm.allgather / comm.Barrier of the control data around a fit
//                                            fitsnap3lib/parallel_tools.py:245-249, 426-441, 562-592
//
// Why a second transport.  The payload of a fit is K^2 + K + 3 doubles -- 132 KB at K = 128, 1.8 MB as a triangle at
// K = 480: latency-bound on any fabric.  A ring or tree collective pays one hop per step; xGMI is point-to-point between
// all GPUs of a node, so every GPU can read every peer's buffer directly: ONE launch per fit in which every rank copies
// its statistics into its own window, raises a flag in every peer's window, waits for the peers' flags and sums the N
// windows IN RANK ORDER into its own statistics buffer ("a hand-rolled one-shot -- every GPU reads its
// peers' buffers via xGMI P2P and sums locally").  The same order on every rank makes the sums bit-identical everywhere,
// which is what lets every rank solve for itself without a broadcast.  And hipIpc handles open between two processes
// that share ONE device, so the N > 1 code paths run on a one-GPU box (RCCL refuses two ranks on one device).
//
// Protocol of all-reduce number g (slot s = g & 1 of a double-buffered window):
//   1. workgroup b copies piece b of the rank's buffer into window slot s, fences at system scope and stores g into
//      flag[s][b][me] of EVERY rank's window (a push: polling then stays in local memory);
//   2. one lane per peer polls flag[s][b][p] >= g in its own window (bounded: s_sleep + wall clock, a status word in
//      page-locked memory and NaN results instead of a hang) -- piece b depends on piece b of the peers only;
//   3. buffer[i] = slot_0[i] + slot_1[i] + ... + slot_{N-1}[i] over piece b.
// Slot s is written again at g + 2: by then this rank has passed step 2 of g + 1, i.e. every peer has published g + 1,
// which it does only after its kernel of g (stream order) has finished reading.  No further handshake.
//
// Host-side collectives (broadcast of control data, all-gather of error tables, the scalar reductions): mailboxes in the
// same shared-memory segment that carried the IPC handles -- no GPU launch, no staging copy; every wait bounded.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>

#include "fsnap_ctx.h"
#include "fsnap_kernels.h"
#include "fsnap_p2p.h"

namespace fsnap {

namespace {

constexpr int P2P_MAX_RANKS = 16;
constexpr char P2P_MAGIC[8] = {'F', 'S', 'N', 'P', '2', 'P', '0', '1'};
constexpr int P2P_MAX_BLOCKS = 128;
constexpr size_t P2P_FLAG_BYTES = 2 * (size_t)P2P_MAX_BLOCKS * P2P_MAX_RANKS * 8;       // head of a window: flag[2][workgroup][rank] (uint64)

// ---- shared-memory segment ------------------------------------------------------------------------------------------
struct ShmRank {
    std::atomic<uint64_t> joined;     // 1: handle / pid / device below are valid
    std::atomic<uint64_t> opened;     // 1: this rank has mapped every peer's window
    std::atomic<uint64_t> posted;     // host collectives: sequence number whose mailbox content is complete
    std::atomic<uint64_t> done;       // ... and the last sequence number this rank has finished READING
    std::atomic<uint64_t> leaving;    // 1: the rank is tearing its communicator down
    hipIpcMemHandle_t handle;
    int64_t pid;
    int32_t device;
    uint64_t raw;                     // the window's address in its owner's process (ranks that share a process)
    char pad[64];
};

struct ShmHeader {
    char magic[8];
    int32_t nranks;
    int32_t pad0;
    uint64_t mailbox_bytes;
    uint64_t slot_bytes;
    ShmRank rank[P2P_MAX_RANKS];
};

size_t shm_bytes(int nranks, size_t mailbox_bytes) { return sizeof(ShmHeader) + (size_t)nranks * 2 * mailbox_bytes; }

size_t env_mb(const char* name, size_t dflt_mb) {
    const char* e = getenv(name);
    const long v = e && *e ? atol(e) : 0;
    return (size_t)(v > 0 ? v : (long)dflt_mb) << 20;
}

}  // namespace

struct P2P {
    int nranks = 1, rank = 0;
    // host segment
    ShmHeader* shm = nullptr;
    size_t shm_len = 0;
    std::string shm_name;
    size_t mailbox_bytes = 0;
    uint64_t hseq = 0;
    // device window: [flags | slot 0 | slot 1]
    char* win = nullptr;
    size_t slot_bytes = 0;
    char* peer[P2P_MAX_RANKS] = {};
    bool peer_ipc[P2P_MAX_RANKS] = {};
    uint64_t gen = 0;
    int* h_status = nullptr;          // page-locked, device-visible: set by a kernel whose wait ran out
};

// ---- the kernel -----------------------------------------------------------------------------------------------------
struct P2PArgs {
    double* slot[P2P_MAX_RANKS];              // slot s of every rank's window, in THIS process's address space
    unsigned long long* flags[P2P_MAX_RANKS]; // flag[s][workgroup][rank] of every rank's window
    int nranks, me;
};

// Workgroup b of EVERY rank owns the same contiguous piece b of the payload (same n, same grid everywhere), so the only
// dependency is "piece b of rank p is in p's window": one flag per (workgroup, rank), no grid-wide ticket -- with one the
// last of 128 workgroups to draw published for all, and the serialised draws cost ~40 us of a 54 us all-reduce at K = 480.
__global__ __launch_bounds__(256) void fsnap_p2p_allreduce_k(double* __restrict__ buf, long long n, long long piece, P2PArgs a,
                                                             unsigned long long gen, int* __restrict__ status,
                                                             unsigned long long timeout_ticks) {
    const long long lo = (long long)blockIdx.x * piece, hi = lo + piece < n ? lo + piece : n;     // piece is even: lo is 16-byte aligned
    const long long cnt = hi - lo, cnt2 = cnt >> 1;
    // 1. own statistics -> own window (16-byte accesses; an odd last element by thread 0)
    {
        const double2* __restrict__ src = reinterpret_cast<const double2*>(buf + lo);
        double2* __restrict__ dst = reinterpret_cast<double2*>(a.slot[a.me] + lo);
        for (long long i = threadIdx.x; i < cnt2; i += 256) dst[i] = src[i];
        if (threadIdx.x == 0 && (cnt & 1)) a.slot[a.me][hi - 1] = buf[hi - 1];
    }
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __threadfence_system();
    __syncthreads();
    // 2. publish piece b in every rank's window, then wait for every rank's piece b in the LOCAL window
    if ((int)threadIdx.x < a.nranks) {
        __hip_atomic_store(a.flags[threadIdx.x] + (size_t)blockIdx.x * P2P_MAX_RANKS + a.me, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* f = a.flags[a.me] + (size_t)blockIdx.x * P2P_MAX_RANKS + threadIdx.x;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gen) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0 && wall_clock64() - t0 > timeout_ticks) {
                s_bad = 1;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // the peers' slots were written by other agents: nothing cached may serve them
    if (s_bad) {
        if (threadIdx.x == 0) *status = 1;
        const double nan = __builtin_nan("");
        for (long long i = lo + threadIdx.x; i < hi; i += 256) buf[i] = nan;
        return;
    }
    // 3. sum in rank order (the same order on every rank: bit-identical results)
    {
        double2* __restrict__ dst = reinterpret_cast<double2*>(buf + lo);
        for (long long i = threadIdx.x; i < cnt2; i += 256) {
            double2 acc = reinterpret_cast<const double2*>(a.slot[0] + lo)[i];
            for (int p = 1; p < a.nranks; ++p) {
                const double2 v = reinterpret_cast<const double2*>(a.slot[p] + lo)[i];
                acc.x += v.x;
                acc.y += v.y;
            }
            dst[i] = acc;
        }
        if (threadIdx.x == 0 && (cnt & 1)) {
            double acc = a.slot[0][hi - 1];
            for (int p = 1; p < a.nranks; ++p) acc += a.slot[p][hi - 1];
            buf[hi - 1] = acc;
        }
    }
}

// ---- ids ------------------------------------------------------------------------------------------------------------
int p2p_make_id(char* id) {
    memset(id, 0, FSNAP_COMM_ID_BYTES);
    memcpy(id, P2P_MAGIC, 8);
    unsigned char rnd[24];
    bool have = false;
    if (FILE* f = fopen("/dev/urandom", "rb")) {
        have = fread(rnd, 1, sizeof rnd, f) == sizeof rnd;
        fclose(f);
    }
    if (!have) {
        uint64_t x = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32);
        for (size_t i = 0; i < sizeof rnd; ++i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            rnd[i] = (unsigned char)(x >> 56);
        }
    }
    memcpy(id + 8, rnd, sizeof rnd);
    return FSNAP_OK;
}

bool p2p_is_id(const char* id) { return memcmp(id, P2P_MAGIC, 8) == 0; }

// ---- bounded host waits ---------------------------------------------------------------------------------------------
namespace {

template <class Pred>
bool spin_until(Pred&& ok, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (ok()) return true;
        if ((spins & 63u) == 63u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
            if (spins > 20000u) std::this_thread::yield();          // long waits give the core away (two ranks may share it)
        }
    }
}

int p2p_timeout(fsnap_ctx* ctx, P2P* p, const char* what) {
    ctx->comm_broken = true;
    return ctx->fail(FSNAP_E_HIP,
                     "rank %d of %d: %s did not finish within %.0f s (FSNAP_COMM_TIMEOUT / option comm_timeout): a peer rank died or "
                     "never reached it",
                     p->rank, p->nranks, what, comm_timeout_s(ctx));
}

char* mailbox(P2P* p, int rank, uint64_t seq) {
    return reinterpret_cast<char*>(p->shm) + sizeof(ShmHeader) + ((size_t)rank * 2 + (seq & 1)) * p->mailbox_bytes;
}

// One round of "everybody posts up to mailbox_bytes, everybody reads what it needs".  post == nullptr: this rank has
// nothing to say in this round (it still takes part in the sequence numbers).
template <class Reader>
int mailbox_round(fsnap_ctx* ctx, P2P* p, const void* post, size_t nbytes, Reader&& reader, const char* what) {
    const uint64_t seq = ++p->hseq;
    const double tmo = comm_timeout_s(ctx);
    ShmHeader* h = p->shm;
    // the slot of this parity was last used by round seq - 2: every rank must have finished reading it
    if (seq > 2 && !spin_until([&] {
            for (int q = 0; q < p->nranks; ++q)
                if (h->rank[q].done.load(std::memory_order_acquire) < seq - 2) return false;
            return true;
        }, tmo))
        return p2p_timeout(ctx, p, what);
    if (post && nbytes) memcpy(mailbox(p, p->rank, seq), post, nbytes);
    h->rank[p->rank].posted.store(seq, std::memory_order_release);
    for (int q = 0; q < p->nranks; ++q) {
        if (!spin_until([&] { return h->rank[q].posted.load(std::memory_order_acquire) >= seq; }, tmo)) return p2p_timeout(ctx, p, what);
        reader(q, mailbox(p, q, seq));
    }
    h->rank[p->rank].done.store(seq, std::memory_order_release);
    return FSNAP_OK;
}

}  // namespace

// ---- init / destroy -------------------------------------------------------------------------------------------------
int p2p_init(fsnap_ctx* ctx, int nranks, int rank, const char* id, P2P** out) {
    *out = nullptr;
    if (nranks > P2P_MAX_RANKS)
        return ctx->fail(FSNAP_E_ARG, "the peer-to-peer transport serves the GPUs of one node: at most %d ranks (got %d)", P2P_MAX_RANKS, nranks);
    P2P* p = new (std::nothrow) P2P();
    if (!p) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
    p->nranks = nranks;
    p->rank = rank;
    p->mailbox_bytes = env_mb("FSNAP_P2P_MAILBOX_MB", 4);
    p->slot_bytes = env_mb("FSNAP_P2P_SLOT_MB", 24);
    auto bail = [&](int rc) {
        p2p_destroy(ctx, p, true);
        return rc;
    };
    // 1. the shared-memory segment, named after the id
    char name[64];
    {
        static const char hex[] = "0123456789abcdef";
        char* w = name + snprintf(name, sizeof name, "/fsnap_p2p_");
        for (int i = 8; i < 24; ++i) {
            *w++ = hex[((unsigned char)id[i]) >> 4];
            *w++ = hex[((unsigned char)id[i]) & 15];
        }
        *w = 0;
    }
    p->shm_name = name;
    p->shm_len = shm_bytes(nranks, p->mailbox_bytes);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return bail(ctx->fail(FSNAP_E_HIP, "shm_open(%s) failed: %s", name, strerror(errno)));
    if (ftruncate(fd, (off_t)p->shm_len) != 0) {
        close(fd);
        return bail(ctx->fail(FSNAP_E_NOMEM, "ftruncate(%s, %zu) failed: %s", name, p->shm_len, strerror(errno)));
    }
    void* m = mmap(nullptr, p->shm_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return bail(ctx->fail(FSNAP_E_NOMEM, "mmap(%s) failed: %s", name, strerror(errno)));
    p->shm = static_cast<ShmHeader*>(m);
    ShmHeader* h = p->shm;
    // 2. this rank's window + what its kernels need
    const size_t win_bytes = P2P_FLAG_BYTES + 2 * p->slot_bytes;
    // fine-grained device memory: a peer GPU reads it coherently inside a running kernel (FSNAP_P2P_WINDOW=coarse: plain
    // hipMalloc memory, for runtimes that cannot export or map the former)
    const char* wkind = getenv("FSNAP_P2P_WINDOW");
    hipError_t e = (wkind && !strcmp(wkind, "coarse")) ? hipErrorUnknown
                                                       : hipExtMallocWithFlags((void**)&p->win, win_bytes, hipDeviceMallocFinegrained);
    hipIpcMemHandle_t handle;
    if (e == hipSuccess && hipIpcGetMemHandle(&handle, p->win) != hipSuccess) {      // (a runtime that cannot export fine-grained memory)
        (void)hipGetLastError();
        (void)hipFree(p->win);
        p->win = nullptr;
        e = hipErrorUnknown;
    }
    const bool fine = e == hipSuccess;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if ((e = hipMalloc((void**)&p->win, win_bytes)) != hipSuccess) return bail(ctx->hipfail(e, "hipMalloc(p2p window)"));
        if ((e = hipIpcGetMemHandle(&handle, p->win)) != hipSuccess)
            return bail(ctx->hipfail(e, "hipIpcGetMemHandle (multi-process GPU sharing needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver)"));
    }
    if (getenv("FSNAP_P2P_DEBUG"))
        fprintf(stderr, "[fsnap p2p] rank %d of %d on device %d: window of %zu bytes, %s device memory\n", rank, nranks, ctx->device, win_bytes,
                fine ? "fine-grained" : "coarse-grained (hipMalloc)");
    if ((e = hipMemset(p->win, 0, P2P_FLAG_BYTES)) != hipSuccess) return bail(ctx->hipfail(e, "hipMemset(p2p flags)"));
    if ((e = hipHostMalloc((void**)&p->h_status, sizeof(int), hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess)
        return bail(ctx->hipfail(e, "hipHostMalloc(p2p status)"));
    *p->h_status = 0;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(ctx->hipfail(e, "hipDeviceSynchronize"));
    // 3. publish, wait for everybody, map the peers
    ShmRank& mine = h->rank[rank];
    mine.handle = handle;
    mine.pid = (int64_t)getpid();
    mine.device = ctx->device;
    mine.raw = (uint64_t)(uintptr_t)p->win;
    if (rank == 0) {
        h->nranks = nranks;
        h->mailbox_bytes = p->mailbox_bytes;
        h->slot_bytes = p->slot_bytes;
        memcpy(h->magic, P2P_MAGIC, 8);
    }
    mine.joined.store(1, std::memory_order_release);
    const double tmo = comm_timeout_s(ctx);
    if (!spin_until([&] {
            for (int q = 0; q < nranks; ++q)
                if (h->rank[q].joined.load(std::memory_order_acquire) != 1) return false;
            return true;
        }, tmo)) {
        if (rank == 0) shm_unlink(name);
        return bail(ctx->fail(FSNAP_E_HIP,
                              "peer-to-peer rendezvous: rank %d of %d did not see every rank within %.0f s (FSNAP_COMM_TIMEOUT): a rank is "
                              "missing or holds a different communicator id",
                              rank, nranks, tmo));
    }
    if (h->mailbox_bytes != p->mailbox_bytes || h->slot_bytes != p->slot_bytes) {
        if (rank == 0) shm_unlink(name);
        return bail(ctx->fail(FSNAP_E_ARG, "peer-to-peer rendezvous: FSNAP_P2P_MAILBOX_MB / FSNAP_P2P_SLOT_MB differ between the ranks"));
    }
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) {
            p->peer[q] = p->win;
        } else if (h->rank[q].pid == mine.pid) {
            p->peer[q] = (char*)(uintptr_t)h->rank[q].raw;          // two contexts of one process: the pointer itself
        } else {
            if (h->rank[q].device != ctx->device) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, ctx->device, h->rank[q].device);
                if (can && hipDeviceEnablePeerAccess(h->rank[q].device, 0) != hipSuccess) (void)hipGetLastError();   // (already enabled)
            }
            void* ptr = nullptr;
            if ((e = hipIpcOpenMemHandle(&ptr, h->rank[q].handle, hipIpcMemLazyEnablePeerAccess)) != hipSuccess) {
                if (rank == 0) shm_unlink(name);
                return bail(ctx->hipfail(e, "hipIpcOpenMemHandle(peer window)"));
            }
            p->peer[q] = (char*)ptr;
            p->peer_ipc[q] = true;
        }
    }
    mine.opened.store(1, std::memory_order_release);
    const bool all_open = spin_until([&] {
        for (int q = 0; q < nranks; ++q)
            if (h->rank[q].opened.load(std::memory_order_acquire) != 1) return false;
        return true;
    }, tmo);
    if (rank == 0) shm_unlink(name);          // the mappings stay; the name is no longer needed (and cannot leak)
    if (!all_open) return bail(ctx->fail(FSNAP_E_HIP, "peer-to-peer rendezvous: a rank failed to map its peers' windows"));
    *out = p;
    return FSNAP_OK;
}

void p2p_destroy(fsnap_ctx* ctx, P2P* p, bool broken) {
    if (!p) return;
    if (p->shm && p->shm->rank[p->rank].opened.load() == 1) {
        // a peer may still be reading this rank's window (its last kernel) or its mailbox: say good-bye and give the others
        // a moment to do the same -- bounded and short: a dead peer must not keep this rank
        ShmHeader* h = p->shm;
        h->rank[p->rank].leaving.store(1, std::memory_order_release);
        if (!broken)
            (void)spin_until([&] {
                for (int q = 0; q < p->nranks; ++q)
                    if (h->rank[q].leaving.load(std::memory_order_acquire) != 1) return false;
                return true;
            }, comm_timeout_s(ctx) < 10.0 ? comm_timeout_s(ctx) : 10.0);
    }
    for (int q = 0; q < p->nranks; ++q)
        if (p->peer_ipc[q] && p->peer[q]) (void)hipIpcCloseMemHandle(p->peer[q]);
    if (p->win) (void)hipFree(p->win);
    if (p->h_status) (void)hipHostFree(p->h_status);
    if (p->shm) munmap(p->shm, p->shm_len);
    delete p;
}

bool p2p_failed(const P2P* p) { return p && p->h_status && *(volatile int*)p->h_status != 0; }

// ---- collectives ----------------------------------------------------------------------------------------------------
int p2p_allreduce_device(fsnap_ctx* ctx, P2P* p, double* d_buf, int64_t n) {
    const int64_t cap = (int64_t)(p->slot_bytes / 8);
    const double tmo = comm_timeout_s(ctx);
    for (int64_t off = 0; off < n; off += cap) {
        const int64_t len = n - off < cap ? n - off : cap;
        const uint64_t gen = ++p->gen;
        P2PArgs a;
        a.nranks = p->nranks;
        a.me = p->rank;
        for (int q = 0; q < p->nranks; ++q) {
            a.slot[q] = reinterpret_cast<double*>(p->peer[q] + P2P_FLAG_BYTES + (gen & 1) * p->slot_bytes);
            a.flags[q] = reinterpret_cast<unsigned long long*>(p->peer[q]) + (gen & 1) * (size_t)P2P_MAX_BLOCKS * P2P_MAX_RANKS;
        }
        // a latency-bound launch: ~2 KiB per thread-block pass, at most 128 workgroups -- few enough to be resident at once
        // beside whatever else runs (every workgroup polls); the same grid on every rank (it follows from len alone)
        int64_t blocks = (len + 2047) / 2048;
        if (blocks < 1) blocks = 1;
        if (blocks > P2P_MAX_BLOCKS) blocks = P2P_MAX_BLOCKS;
        int64_t piece = (len + blocks - 1) / blocks;
        piece += piece & 1;
        blocks = (len + piece - 1) / piece;
        hipLaunchKernelGGL(fsnap_p2p_allreduce_k, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_buf + off, (long long)len,
                           (long long)piece, a, (unsigned long long)gen, p->h_status, (unsigned long long)(tmo * 1.0e8));
        FSNAP_HIP(hipGetLastError(), "launch fsnap_p2p_allreduce_k");
    }
    return FSNAP_OK;
}

int p2p_allgather_host(fsnap_ctx* ctx, P2P* p, const void* send, size_t nbytes, void* recv) {
    const char* s = static_cast<const char*>(send);
    char* r = static_cast<char*>(recv);
    for (size_t off = 0; off < nbytes; off += p->mailbox_bytes) {
        const size_t len = nbytes - off < p->mailbox_bytes ? nbytes - off : p->mailbox_bytes;
        const int rc = mailbox_round(ctx, p, s + off, len, [&](int q, const char* box) { memcpy(r + (size_t)q * nbytes + off, box, len); },
                                     "all-gather");
        if (rc) return rc;
    }
    return FSNAP_OK;
}

int p2p_bcast_host(fsnap_ctx* ctx, P2P* p, void* buf, size_t nbytes, int root) {
    char* b = static_cast<char*>(buf);
    for (size_t off = 0; off < nbytes; off += p->mailbox_bytes) {
        const size_t len = nbytes - off < p->mailbox_bytes ? nbytes - off : p->mailbox_bytes;
        const int rc = mailbox_round(ctx, p, p->rank == root ? b + off : nullptr, len,
                                     [&](int q, const char* box) {
                                         if (q == root && p->rank != root) memcpy(b + off, box, len);
                                     },
                                     "broadcast");
        if (rc) return rc;
    }
    return FSNAP_OK;
}

int p2p_allreduce_host(fsnap_ctx* ctx, P2P* p, double* buf, int64_t n, int op) {
    const size_t cap = p->mailbox_bytes / 8;
    for (int64_t off = 0; off < n; off += (int64_t)cap) {
        const size_t len = (size_t)(n - off) < cap ? (size_t)(n - off) : cap;
        double* out = buf + off;
        bool first = true;
        // rank order: the same association on every rank
        const int rc = mailbox_round(ctx, p, out, len * 8,
                                     [&](int, const char* box) {
                                         const double* v = reinterpret_cast<const double*>(box);
                                         if (first) {
                                             for (size_t i = 0; i < len; ++i) out[i] = v[i];
                                             first = false;
                                         } else if (op == 0) {
                                             for (size_t i = 0; i < len; ++i) out[i] += v[i];
                                         } else if (op == 1) {
                                             for (size_t i = 0; i < len; ++i) out[i] = v[i] > out[i] || v[i] != v[i] ? v[i] : out[i];
                                         } else {
                                             for (size_t i = 0; i < len; ++i) out[i] = v[i] < out[i] || v[i] != v[i] ? v[i] : out[i];
                                         }
                                     },
                                     "all-reduce");
        if (rc) return rc;
    }
    return FSNAP_OK;
}

int p2p_barrier(fsnap_ctx* ctx, P2P* p) {
    return mailbox_round(ctx, p, nullptr, 0, [](int, const char*) {}, "barrier");
}

}  // namespace fsnap
