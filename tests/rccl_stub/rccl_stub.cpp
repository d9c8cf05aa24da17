// Stand-in for librccl in the CPU test tier (TEST INFRASTRUCTURE ONLY, never loaded by the product).
//
// csrc/comm.inl binds six RCCL entry points with dlopen/dlsym (PEPPA_RCCL_LIBRARY selects the library).  This file implements
// exactly those six over a POSIX shared-memory segment so that tests/test_comm_two_ranks.py can drive pf_comm_unique_id /
// pf_broadcast_weights from TWO processes (rank 0 and rank 1) against the SIMT-emulator flavour of the engine, whose "device
// memory" is host memory: the rank != 0 branch of pf_broadcast_weights (receive-capacity check, copy into the caller's buffer,
// program load on the receiving rank) runs for real, only the transport is faked.
//
// Protocol: the unique id carries the segment's name.  A communicator is (segment, rank, world).  A broadcast is a sequence of
// chunk rounds; in each round the root copies a chunk into the segment and publishes the round number, every other rank copies
// it out and acknowledges, the root waits for world-1 acknowledgements before the next round.  Collectives are matched by call
// order per communicator, as with the real library; a rank that never arrives makes the others time out with an error instead
// of hanging the test suite.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {

const size_t kChunk = 4u << 20;
double timeout_s() {
    const char* e = getenv("PF_RCCL_STUB_TIMEOUT");      // seconds; tests of the failure paths shorten it
    return e && *e ? atof(e) : 60.0;
}

struct Segment {
    std::atomic<int> joined;          // ranks that have called ncclCommInitRank
    std::atomic<int> world;           // world size announced by the first rank (others must agree)
    std::atomic<uint64_t> round;      // number of the chunk round the root has published
    std::atomic<uint64_t> acks;       // acknowledgements of the published round
    std::atomic<uint64_t> chunk_bytes;
    unsigned char data[kChunk];
};

struct Comm {
    Segment* seg;
    int rank, world;
    uint64_t next_round;              // this rank's view of the collective sequence
    char name[128];
};

struct Uid { char internal[128]; };

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

template <typename F> bool wait_until(F cond) {
    const double t0 = now_s(), limit = timeout_s();
    while (!cond()) {
        if (now_s() - t0 > limit) return false;
        usleep(50);
    }
    return true;
}

enum { kOk = 0, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4 };

}  // namespace

extern "C" {

int ncclGetVersion(int* v) { if (!v) return kInvalidArgument; *v = 99900; return kOk; }   // "stub" version

const char* ncclGetErrorString(int r) {
    switch (r) {
        case kOk: return "no error";
        case kSystemError: return "rccl stub: shared-memory segment could not be opened";
        case kInternalError: return "rccl stub: a rank did not arrive within the time-out";
        case kInvalidArgument: return "rccl stub: invalid argument";
        default: return "rccl stub: unknown error";
    }
}

int ncclGetUniqueId(Uid* id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof(*id));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/pf_rccl_stub_%d_%lx", (int)getpid(), (unsigned long)ts.tv_nsec);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    if (ftruncate(fd, sizeof(Segment)) != 0) { close(fd); shm_unlink(id->internal); return kSystemError; }
    close(fd);                                   // zero-filled: joined = world = round = acks = 0
    return kOk;
}

int ncclCommInitRank(void** comm, int world, Uid id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return kInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return kSystemError;
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return kSystemError;
    Segment* seg = static_cast<Segment*>(p);
    int expected = 0;
    if (!seg->world.compare_exchange_strong(expected, world) && expected != world) { munmap(p, sizeof(Segment)); return kInvalidArgument; }
    seg->joined.fetch_add(1);
    if (!wait_until([&] { return seg->joined.load() >= world; })) { munmap(p, sizeof(Segment)); return kInternalError; }
    Comm* c = new Comm();
    c->seg = seg; c->rank = rank; c->world = world; c->next_round = 1;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    *comm = c;
    return kOk;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, void* comm, void* /*stream*/) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || dtype != 1 /* ncclUint8 */ || root < 0 || root >= c->world) return kInvalidArgument;
    Segment* seg = c->seg;
    size_t done = 0;
    do {                                           // at least one round, so that a zero-byte broadcast still synchronises
        const size_t n = count - done < kChunk ? count - done : kChunk;
        const uint64_t r = c->next_round++;
        if (c->rank == root) {
            if (!wait_until([&] { return seg->round.load() == r - 1 && (r == 1 || seg->acks.load() == (uint64_t)(c->world - 1)); })) return kInternalError;
            memcpy(seg->data, static_cast<const unsigned char*>(send) + done, n);
            seg->chunk_bytes.store(n);
            seg->acks.store(0);
            seg->round.store(r);
            if (recv != send) memcpy(static_cast<unsigned char*>(recv) + done, static_cast<const unsigned char*>(send) + done, n);
            if (!wait_until([&] { return seg->acks.load() == (uint64_t)(c->world - 1); })) return kInternalError;
        } else {
            if (!wait_until([&] { return seg->round.load() == r; })) return kInternalError;
            if (seg->chunk_bytes.load() != n) return kInvalidArgument;       // ranks disagree about the size of the collective
            memcpy(static_cast<unsigned char*>(recv) + done, seg->data, n);
            seg->acks.fetch_add(1);
        }
        done += n;
    } while (done < count);
    return kOk;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return kInvalidArgument;
    if (c->seg->joined.fetch_sub(1) == 1) shm_unlink(c->name);     // the last rank out removes the segment
    munmap(c->seg, sizeof(Segment));
    delete c;
    return kOk;
}

}  // extern "C"
