// comm_rccl.cpp -- the RCCL transport behind kfx_comm (include/kfx_slab.h): one process per GPU, all-reduce and grouped
// neighbour send / recv over xGMI.  Built into libkfx_rccl.so so that libkfx.so itself does not depend on librccl.
// The ncclUniqueId travels through a file: rank 0 removes whatever is there, creates the file atomically (exclusive
// temporary + rename) with a launch nonce in front of the id, the others wait for a file that is a regular file of this user,
// carries their nonce and was not written before their own process began, give or take the launch skew between ranks
// (rdv_slack_s(): 10 s, KFX_RDV_SLACK_S overrides; process_start() is the real start of the process, so a rank that spends a
// minute importing still accepts the file rank 0 published meanwhile).  The nonce folds in the PARENT process (pid + its start
// time) beside the launcher's environment: the ranks of one launch are children of one launcher, a relaunch has another -- a
// file left behind by a crashed run with the same MASTER_PORT is refused whatever its age (round-4 advice: torchrun's default
// port made the environment part static across relaunches).  KFX_RDV_PARENT=0 leaves the parent out (ranks started by hand from
// different shells: give them a KFX_RUN_ID that is unique per launch instead).
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/kfx_slab.h"

namespace {

struct RcclImpl {
    ncclComm_t comm = nullptr;
    void* token = nullptr; // 4-byte device word for the barrier
};

int nccl_status(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }

int rccl_all_reduce(kfx_comm* c, void* buf, size_t count, int op, kfx_stream stream)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    if (count == 0) return 0;
    if (!buf) return KFX_E_NULL;
    ncclDataType_t dt;
    ncclRedOp_t ro;
    switch (op) {
    case KFX_COMM_MIN_I64: dt = ncclInt64; ro = ncclMin; break;
    case KFX_COMM_SUM_F32: dt = ncclFloat32; ro = ncclSum; break;
    case KFX_COMM_SUM_I32: dt = ncclInt32; ro = ncclSum; break;
    default: return KFX_E_RANGE;
    }
    return nccl_status(ncclAllReduce(buf, buf, count, dt, ro, im->comm, (hipStream_t)stream));
}

// one batched group of point-to-point operations with the two neighbour ranks; every leg has its own size (0: skipped)
int rccl_exchange_v(kfx_comm* c, const void* send_lo, size_t bytes_send_lo, void* recv_lo, size_t bytes_recv_lo, const void* send_hi, size_t bytes_send_hi,
                    void* recv_hi, size_t bytes_recv_hi, kfx_stream stream)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    hipStream_t s = (hipStream_t)stream;
    const bool has_lo = c->rank > 0, has_hi = c->rank + 1 < c->world;
    const bool sl = has_lo && bytes_send_lo, rl = has_lo && bytes_recv_lo, sh = has_hi && bytes_send_hi, rh = has_hi && bytes_recv_hi;
    if (!sl && !rl && !sh && !rh) return 0;
    if ((sl && !send_lo) || (rl && !recv_lo) || (sh && !send_hi) || (rh && !recv_hi)) return KFX_E_NULL;
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess && sl) r = ncclSend(send_lo, bytes_send_lo, ncclInt8, c->rank - 1, im->comm, s);
    if (r == ncclSuccess && rl) r = ncclRecv(recv_lo, bytes_recv_lo, ncclInt8, c->rank - 1, im->comm, s);
    if (r == ncclSuccess && sh) r = ncclSend(send_hi, bytes_send_hi, ncclInt8, c->rank + 1, im->comm, s);
    if (r == ncclSuccess && rh) r = ncclRecv(recv_hi, bytes_recv_hi, ncclInt8, c->rank + 1, im->comm, s);
    const ncclResult_t e = ncclGroupEnd();
    return nccl_status(r != ncclSuccess ? r : e);
}

int rccl_exchange(kfx_comm* c, const void* send_lo, void* recv_lo, size_t bytes_lo, const void* send_hi, void* recv_hi, size_t bytes_hi,
                  kfx_stream stream)
{
    return rccl_exchange_v(c, send_lo, bytes_lo, recv_lo, bytes_lo, send_hi, bytes_hi, recv_hi, bytes_hi, stream);
}

int rccl_broadcast(kfx_comm* c, void* buf, size_t bytes, int root, kfx_stream stream)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    if (bytes == 0) return 0;
    if (!buf) return KFX_E_NULL;
    if (root < 0 || root >= c->world) return KFX_E_RANGE;
    return nccl_status(ncclBroadcast(buf, buf, bytes, ncclInt8, root, im->comm, (hipStream_t)stream));
}

// One grouped send / recv per peer: RCCL runs them concurrently, each pair over its own xGMI link
int rccl_all_to_all(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    if (bytes == 0) return 0;
    if (!send || !recv) return KFX_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    ncclResult_t r = ncclGroupStart();
    for (int k = 0; k < c->world && r == ncclSuccess; ++k) {
        const int peer = (c->rank + k) % c->world;   // (the own chunk too: a device-local copy inside the group)
        r = ncclSend(static_cast<const char*>(send) + (size_t)peer * bytes, bytes, ncclInt8, peer, im->comm, s);
        if (r == ncclSuccess) r = ncclRecv(static_cast<char*>(recv) + (size_t)peer * bytes, bytes, ncclInt8, peer, im->comm, s);
    }
    const ncclResult_t e = ncclGroupEnd();
    return nccl_status(r != ncclSuccess ? r : e);
}

int rccl_all_gather(kfx_comm* c, const void* send, void* recv, size_t bytes, kfx_stream stream)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    if (bytes == 0) return 0;
    if (!send || !recv) return KFX_E_NULL;
    return nccl_status(ncclAllGather(send, recv, bytes, ncclInt8, im->comm, (hipStream_t)stream));
}

int rccl_barrier(kfx_comm* c)
{
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    const ncclResult_t r = ncclAllReduce(im->token, im->token, 1, ncclInt32, ncclSum, im->comm, nullptr);
    if (r != ncclSuccess) return nccl_status(r);
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : 1;
}

void rccl_destroy(kfx_comm* c)
{
    if (!c || !c->impl) return;
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    if (im->comm) ncclCommDestroy(im->comm);
    if (im->token) (void)hipFree(im->token);
    delete im;
    c->impl = nullptr;
}

int rccl_fill(kfx_comm* comm, RcclImpl* im, int rank, int world);

// A second communicator over the same ranks (ncclCommSplit, one colour, the same rank order): its operations are matched among
// themselves only, so a stream that issues through it needs no common order with the streams that use the original.  Every rank calls.
int rccl_dup(kfx_comm* c, kfx_comm* out)
{
    if (!c || !out || !c->impl) return KFX_E_NULL;
    RcclImpl* im = static_cast<RcclImpl*>(c->impl);
    RcclImpl* n = new (std::nothrow) RcclImpl;
    if (!n) return KFX_E_RANGE;
    const ncclResult_t r = ncclCommSplit(im->comm, 0, c->rank, &n->comm, nullptr);
    if (r != ncclSuccess || !n->comm) {
        delete n;
        return nccl_status(r != ncclSuccess ? r : ncclInternalError);
    }
    if (hipMalloc(&n->token, 4) != hipSuccess || hipMemset(n->token, 0, 4) != hipSuccess) {
        ncclCommDestroy(n->comm);
        delete n;
        return KFX_E_NODEVICE;
    }
    return rccl_fill(out, n, c->rank, c->world);
}

int rccl_fill(kfx_comm* comm, RcclImpl* im, int rank, int world)
{
    comm->rank = rank;
    comm->world = world;
    comm->impl = im;
    comm->all_reduce = rccl_all_reduce;
    comm->exchange = rccl_exchange;
    comm->barrier = rccl_barrier;
    comm->destroy = rccl_destroy;
    comm->broadcast = rccl_broadcast;
    comm->all_to_all = rccl_all_to_all;
    comm->all_gather = rccl_all_gather;
    comm->exchange_v = rccl_exchange_v;
    comm->flags = 0;   // collectives are enqueued on the stream; the host does not wait for its peers
    comm->dup = rccl_dup;
    return 0;
}

// Field 22 of /proc/<pid>/stat: the start time of a process in clock ticks since boot (-1: unreadable).  The command name in
// field 2 may contain spaces and parentheses, so the fields are counted from the LAST ')'.
long long start_ticks(const char* stat_path)
{
    long long ticks = -1;
    if (FILE* f = fopen(stat_path, "r")) {
        char buf[2048];
        const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[n] = 0;
        if (const char* p = strrchr(buf, ')')) {
            int field = 2;   // p is at the end of field 2; field 3 (the state) follows
            ++p;
            while (*p) {
                while (*p == ' ') ++p;
                if (!*p) break;
                if (++field == 22) { ticks = atoll(p); break; }
                while (*p && *p != ' ') ++p;
            }
        }
    }
    return ticks;
}

// What every rank of one launch has and no rank of another: the launcher's environment values and the launcher itself (parent
// pid + the parent's start time), folded into 64 bits.
uint64_t launch_nonce()
{
    uint64_t h = 1469598103934665603ull;
    const auto fold = [&](const char* v) {
        for (const char* c = v; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        h = (h ^ 0xffu) * 1099511628211ull;
    };
    bool any = false;
    for (const char* name : {"KFX_RUN_ID", "TORCHELASTIC_RUN_ID", "MASTER_PORT", "SLURM_JOB_ID", "SLURM_STEP_ID"}) {
        const char* v = getenv(name);
        if (!v || !*v) continue;
        any = true;
        fold(v);
    }
    // the launcher is part of the nonce unless the launch names itself (KFX_RUN_ID / TORCHELASTIC_RUN_ID are unique per launch: ranks
    // that do not share a parent -- one wrapper shell per rank, a step daemon per task, one agent per node -- still agree) or
    // KFX_RDV_PARENT says which; KFX_RDV_PARENT=1 forces it in, 0 leaves it out
    const char* par = getenv("KFX_RDV_PARENT");
    const char* run_id = getenv("KFX_RUN_ID");
    const char* el_id = getenv("TORCHELASTIC_RUN_ID");
    const bool named = (run_id && *run_id) || (el_id && *el_id && strcmp(el_id, "none") != 0);
    if (par && *par ? atoi(par) != 0 : !named) {
        const long long ppid = (long long)getppid();
        const std::string path = "/proc/" + std::to_string(ppid) + "/stat";
        const std::string id = "ppid:" + std::to_string(ppid) + ":" + std::to_string(start_ticks(path.c_str()));
        any = true;
        fold(id.c_str());
    }
    return any ? (h | 1ull) : 0ull;
}

struct Rendezvous {
    char magic[8];
    uint64_t nonce;
    ncclUniqueId id;
};
const char RDV_MAGIC[8] = {'K', 'F', 'X', 'R', 'D', 'V', '1', 0};

// Start of this process as wall-clock seconds: field 22 of /proc/self/stat (start time in clock ticks since boot; the
// command name in field 2 may contain spaces and parentheses, so the fields are counted from the LAST ')') plus the boot time
// `btime` of /proc/stat.  (The st_mtime of /proc/self is NOT it: procfs stamps that inode when it is first looked up, which
// for a rank that spends seconds importing before it gets here is seconds late -- round-3 advice.)  "now" where procfs is missing.
time_t process_start()
{
    long long btime = -1;
    const long long ticks = start_ticks("/proc/self/stat");
    if (FILE* f = fopen("/proc/stat", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), f))
            if (strncmp(line, "btime ", 6) == 0) { btime = atoll(line + 6); break; }
        fclose(f);
    }
    const long hz = sysconf(_SC_CLK_TCK);
    if (ticks >= 0 && btime > 0 && hz > 0) return (time_t)(btime + ticks / hz);
    return time(nullptr);
}
// A rendezvous file may precede this process by this much and still belong to its launch: the skew between the starts of the
// ranks of one launch (a launcher forks them within milliseconds; a shell loop within seconds), NOT the time a rank takes to get
// here -- process_start() is the real start.  10 s; KFX_RDV_SLACK_S overrides (0 ... 3600).
time_t rdv_slack_s()
{
    static const time_t slack = [] {
        const char* e = getenv("KFX_RDV_SLACK_S");
        const long v = e && *e ? atol(e) : 10;
        return (time_t)(v < 0 ? 0 : (v > 3600 ? 3600 : v));
    }();
    return slack;
}

int write_rendezvous(const char* path, const Rendezvous& rv)
{
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    unlink(path);        // a file of an earlier run must not be readable while this one starts
    unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return KFX_E_RANGE;
    const ssize_t n = write(fd, &rv, sizeof(rv));
    const int c = close(fd);
    if (n != (ssize_t)sizeof(rv) || c != 0 || rename(tmp.c_str(), path) != 0) {
        unlink(tmp.c_str());
        return KFX_E_RANGE;
    }
    return 0;
}

// 1: accepted, 0: not (yet) there / not ours / stale.  *foreign (optional) is set when a well-formed, fresh file of this user was
// seen whose ONLY mismatch is the nonce: ranks of one launch that derive different nonces (no common parent) -- worth saying so
// when the wait times out instead of a generic failure.
int read_rendezvous(const char* path, Rendezvous& rv, uint64_t nonce, time_t not_before, bool* foreign = nullptr)
{
    const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return 0;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && st.st_size == (off_t)sizeof(rv) &&
              st.st_mtime + rdv_slack_s() >= not_before;
    ok = ok && read(fd, &rv, sizeof(rv)) == (ssize_t)sizeof(rv) && memcmp(rv.magic, RDV_MAGIC, 8) == 0;
    close(fd);
    if (ok && rv.nonce != nonce && foreign) *foreign = true;
    return ok && rv.nonce == nonce ? 1 : 0;
}

} // namespace

// Test hook (tests/test_abi_cpu.py; not part of include/kfx_slab.h): would a rank > 0 of this process accept the file at `path`
// right now?  1 / 0.  Reaches no RCCL or HIP call.
extern "C" int kfx_rccl_rendezvous_probe(const char* path)
{
    if (!path) return KFX_E_NULL;
    Rendezvous rv;
    return read_rendezvous(path, rv, launch_nonce(), process_start());
}
extern "C" long long kfx_rccl_process_start(void) { return (long long)process_start(); }
extern "C" unsigned long long kfx_rccl_launch_nonce(void) { return (unsigned long long)launch_nonce(); }

extern "C" int kfx_comm_create_rccl(kfx_comm* comm, int rank, int world, const char* rendezvous_file, int timeout_s)
{
    if (!comm || (world > 1 && !rendezvous_file)) return KFX_E_NULL;
    if (world < 1 || rank < 0 || rank >= world) return KFX_E_RANGE;
    Rendezvous rv;
    memset(&rv, 0, sizeof(rv));
    memcpy(rv.magic, RDV_MAGIC, 8);
    rv.nonce = launch_nonce();
    ncclUniqueId& id = rv.id;
    if (rank == 0) {
        const ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) return nccl_status(r);
        if (world > 1)
            if (int e = write_rendezvous(rendezvous_file, rv)) return e;
    } else {
        const uint64_t nonce = rv.nonce;
        const time_t started = process_start();
        const auto t0 = std::chrono::steady_clock::now();
        bool foreign = false;
        while (!read_rendezvous(rendezvous_file, rv, nonce, started, &foreign)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s > 0 ? timeout_s : 60)) {
                if (foreign)
                    fprintf(stderr, "kfx_comm_create_rccl: rank %d timed out: a rendezvous file is present at %s but its launch nonce differs from this "
                                    "rank's (the ranks do not share a parent process?) -- give every rank the same KFX_RUN_ID, unique per launch "
                                    "(or KFX_RDV_PARENT=0 with a unique MASTER_PORT)\n", rank, rendezvous_file);
                else
                    fprintf(stderr, "kfx_comm_create_rccl: rank %d timed out waiting for rank 0's rendezvous file %s\n", rank, rendezvous_file);
                return KFX_E_RANGE;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    RcclImpl* im = new (std::nothrow) RcclImpl;
    if (!im) return KFX_E_RANGE;
    const ncclResult_t r = ncclCommInitRank(&im->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete im;
        return nccl_status(r);
    }
    if (hipMalloc(&im->token, 4) != hipSuccess || hipMemset(im->token, 0, 4) != hipSuccess) {
        ncclCommDestroy(im->comm);
        delete im;
        return KFX_E_NODEVICE;
    }
    rccl_fill(comm, im, rank, world);
    if (rank == 0 && world > 1) { // every rank has joined once ncclCommInitRank returns: the file has served its purpose
        rccl_barrier(comm);
        unlink(rendezvous_file);
    } else if (world > 1) {
        rccl_barrier(comm);
    }
    return 0;
}
