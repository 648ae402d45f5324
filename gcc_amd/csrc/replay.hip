// Launch replay (include/gcc_hip.h: gcc_replay_*): the models whose iteration is a few thousand small launches are bound by the
// HOST -- ~2.5 us of Python and ~4.3 us of hipLaunchKernel per launch on one thread -- and a HIP graph of the
// iteration replays no faster than the eager path on ROCm 7.2.  This is the runtime's own answer: while a thread records,
// every launch of the library (gcc_launch, common.hpp), its memsets / copies and the event records / waits that order its
// streams are written down with their argument values; gcc_replay_run issues the list again -- no Python, no planning code --
// and, with worker threads, issues every HIP stream's share from its own host thread (the launches of different streams are
// independent on the host; an event wait is held back until the record it saw at recording time has been issued).
//
// The contract is the one of a captured graph: every pointer argument must still be valid and mean the same at replay (the
// caller keeps the iteration's allocations in a private pool and feeds inputs through persistent buffers), and scalars that
// change between iterations are patched by tag (gcc_replay_tag_next / gcc_replay_patch: Adam's step-dependent factors).
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {
enum { E_KERNEL = 0, E_MEMSET, E_MEMCPY, E_RECORD, E_WAIT, E_ALLREDUCE };

struct Entry {
    int type, tag;
    hipStream_t stream;
    const void* func;
    dim3 grid, block;
    unsigned shmem;
    unsigned first_arg, nargs;        // into arg_off / arg_size
    void* dst; const void* src; size_t bytes; int value;      // memset / memcpy
    hipEvent_t ev;                    // the caller's event
    hipEvent_t own;                   // E_RECORD: this record's private event; E_WAIT: the private event of the record it saw
    bool last;                        // E_RECORD: no later record of `ev` in the recording -- the caller's event is recorded as well
    int dep;                          // E_WAIT: index of the E_RECORD entry it saw (-1: recorded before the recording began)
    int lane;                         // worker that issues it
};
}  // namespace

struct gcc_replay {
    std::vector<Entry> e;
    std::vector<unsigned char> blob;              // argument values while recording
    std::vector<unsigned> arg_off, arg_size;
    unsigned char* args = nullptr;                // 64-byte aligned copy of blob (gcc_replay_end)
    std::vector<void*> arg_ptr;                   // per argument: args + arg_off
    bool recording = false, closed = false;
    int pending_tag = 0;
    // threaded issue
    int nlanes = 1;
    std::vector<std::vector<int>> lane_list;      // entry indices per lane, in recorded order
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long long generation = 0;
    int running = 0;
    bool quit = false;
    std::atomic<int>* issued = nullptr;           // per entry: generation in which it was issued (records only are read)
    std::atomic<int> failed{0};
};

namespace {
thread_local gcc_replay* t_rec = nullptr;

int issue(gcc_replay* r, const Entry& x) {
    switch (x.type) {
    case E_KERNEL:
        return hipLaunchKernel(x.func, x.grid, x.block, r->arg_ptr.data() + x.first_arg, x.shmem, x.stream) == hipSuccess ? 0 : 1;
    case E_MEMSET: return hipMemsetAsync(x.dst, x.value, x.bytes, x.stream) == hipSuccess ? 0 : 1;
    case E_MEMCPY: return hipMemcpyAsync(x.dst, x.src, x.bytes, hipMemcpyDeviceToDevice, x.stream) == hipSuccess ? 0 : 1;
    // Every record of the recording has an event of its own: a caller's event that is re-recorded inside the iteration (a
    // side stream's fork event: hundreds of times) would otherwise make a wait depend on WHEN it is issued relative to the
    // later records -- the recording thread issued everything in one order, the replay's threads do not.
    case E_RECORD:
        if (hipEventRecord(x.own, x.stream) != hipSuccess) return 1;
        return (x.last && hipEventRecord(x.ev, x.stream) != hipSuccess) ? 1 : 0;
    case E_WAIT: return hipStreamWaitEvent(x.stream, x.dep >= 0 ? x.own : x.ev, 0) == hipSuccess ? 0 : 1;
    // a gradient all-reduce of the C ABI's communicator (comm.hip): dst = buffer, src = communicator, bytes = count, value = dtype.
    // Collectives of one communicator must be issued in the same order on every rank: a recording that holds one is replayed
    // from ONE thread (gcc_replay_end)
    case E_ALLREDUCE: return gcc_internal_comm_allreduce((void*)x.src, x.dst, x.bytes, x.value, x.stream) == GCC_OK ? 0 : 1;
    }
    return 1;
}

void run_lane(gcc_replay* r, int lane, int gen) {
    for (int idx : r->lane_list[lane]) {
        const Entry& x = r->e[idx];
        if (x.type == E_WAIT && x.dep >= 0 && r->e[x.dep].lane != lane) {
            // the record this wait saw is another thread's to issue: a wait on an event that has not been recorded YET is no wait
            while (r->issued[x.dep].load(std::memory_order_acquire) != gen) {
                if (r->failed.load(std::memory_order_relaxed)) return;
                __builtin_ia32_pause();
            }
        }
        if (issue(r, x)) { r->failed.store(1); return; }
        if (x.type == E_RECORD) r->issued[idx].store(gen, std::memory_order_release);
    }
}

void worker_loop(gcc_replay* r, int lane, int device) {
    (void)hipSetDevice(device);
    long long seen = 0;
    for (;;) {
        int gen;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_go.wait(lk, [&] { return r->quit || r->generation != seen; });
            if (r->quit) return;
            seen = r->generation;
            gen = (int)(seen & 0x3fffffff) + 1;
        }
        run_lane(r, lane, gen);
        {
            std::lock_guard<std::mutex> lk(r->mu);
            if (--r->running == 0) r->cv_done.notify_all();
        }
    }
}
}  // namespace

bool gcc_replay_recording() { return t_rec != nullptr; }
static std::atomic<unsigned> g_generation{0};
unsigned gcc_replay_generation() { return g_generation.load(std::memory_order_relaxed); }

void gcc_replay_record_kernel(const GccLaunchRec& k) {
    gcc_replay* r = t_rec;
    Entry x = {};
    x.type = E_KERNEL; x.tag = r->pending_tag; r->pending_tag = 0;
    x.stream = k.stream; x.func = k.func; x.grid = k.grid; x.block = k.block; x.shmem = k.shmem;
    x.first_arg = (unsigned)r->arg_off.size(); x.nargs = (unsigned)k.nargs; x.dep = -1;
    for (int i = 0; i < k.nargs; i++) {
        const size_t off = (r->blob.size() + 15) & ~(size_t)15;
        r->blob.resize(off + k.sizes[i]);
        memcpy(r->blob.data() + off, k.args[i], k.sizes[i]);
        r->arg_off.push_back((unsigned)off);
        r->arg_size.push_back(k.sizes[i]);
    }
    r->e.push_back(x);
}

hipError_t gcc_memset_async(void* dst, int value, size_t bytes, hipStream_t st) {
    if (t_rec) {
        Entry x = {};
        x.type = E_MEMSET; x.stream = st; x.dst = dst; x.value = value; x.bytes = bytes; x.dep = -1;
        t_rec->e.push_back(x);
    }
    return hipMemsetAsync(dst, value, bytes, st);
}
void gcc_replay_record_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t st) {
    if (!t_rec) return;
    Entry x = {};
    x.type = E_ALLREDUCE; x.stream = st; x.dst = buf; x.src = comm; x.bytes = count; x.value = dtype; x.dep = -1;
    t_rec->e.push_back(x);
}
hipError_t gcc_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (t_rec) {
        Entry x = {};
        x.type = E_MEMCPY; x.stream = st; x.dst = dst; x.src = src; x.bytes = bytes; x.dep = -1;
        t_rec->e.push_back(x);
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
}

// ---- events: the library's own, so that the orderings between the streams of an iteration are part of its recording ----
extern "C" int gcc_event_create(void** ev) {
    GCC_ENTER();
    if (!ev) return GCC_ERR_BAD_ARG;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return GCC_ERR_LAUNCH;
    *ev = (void*)e;
    return GCC_OK;
}
extern "C" int gcc_event_destroy(void* ev) {
    GCC_ENTER();
    if (!ev) return GCC_ERR_BAD_ARG;
    return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? GCC_OK : GCC_ERR_LAUNCH;
}
extern "C" int gcc_event_record(void* ev, gcc_stream_t stream) {
    GCC_ENTER();
    if (!ev) return GCC_ERR_BAD_ARG;
    if (t_rec) {
        Entry x = {};
        x.type = E_RECORD; x.stream = (hipStream_t)stream; x.ev = (hipEvent_t)ev; x.dep = -1;
        t_rec->e.push_back(x);
    }
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? GCC_OK : GCC_ERR_LAUNCH;
}
extern "C" int gcc_stream_wait_event(gcc_stream_t stream, void* ev) {
    GCC_ENTER();
    if (!ev) return GCC_ERR_BAD_ARG;
    if (t_rec) {
        Entry x = {};
        x.type = E_WAIT; x.stream = (hipStream_t)stream; x.ev = (hipEvent_t)ev; x.dep = -1;
        for (int i = (int)t_rec->e.size() - 1; i >= 0; i--)
            if (t_rec->e[i].type == E_RECORD && t_rec->e[i].ev == x.ev) { x.dep = i; break; }
        t_rec->e.push_back(x);
    }
    return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) == hipSuccess ? GCC_OK : GCC_ERR_LAUNCH;
}

// ---- recording ----
extern "C" int gcc_replay_begin(gcc_replay_t** out) {
    if (!out || t_rec) return GCC_ERR_BAD_ARG;          // one recording per thread at a time
    gcc_replay* r = new (std::nothrow) gcc_replay;
    if (!r) return GCC_ERR_LAUNCH;
    r->recording = true;
    g_generation.fetch_add(1, std::memory_order_relaxed);
    t_rec = r;
    *out = r;
    return GCC_OK;
}

// closes the recording of the calling thread.  threads: 0 / 1 = gcc_replay_run issues everything from the calling thread;
// n > 1 = up to n host threads, one per HIP stream (the busiest streams get their own, the rest share the last one)
extern "C" int gcc_replay_end(gcc_replay_t* r, int threads) {
    if (!r || t_rec != r || !r->recording) return GCC_ERR_BAD_ARG;
    t_rec = nullptr;
    r->recording = false; r->closed = true;
    const size_t bytes = (r->blob.size() + 63) & ~(size_t)63;
    r->args = (unsigned char*)aligned_alloc(64, bytes ? bytes : 64);
    if (!r->args) return GCC_ERR_LAUNCH;
    memcpy(r->args, r->blob.data(), r->blob.size());
    r->blob.clear(); r->blob.shrink_to_fit();
    r->arg_ptr.resize(r->arg_off.size());
    for (size_t i = 0; i < r->arg_off.size(); i++) r->arg_ptr[i] = r->args + r->arg_off[i];
    // lanes: streams ordered by their number of entries
    std::vector<hipStream_t> streams;
    std::vector<int> count;
    for (const Entry& x : r->e) {
        size_t k = 0;
        while (k < streams.size() && streams[k] != x.stream) k++;
        if (k == streams.size()) { streams.push_back(x.stream); count.push_back(0); }
        count[k]++;
    }
    std::vector<int> order(streams.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    for (size_t i = 0; i < order.size(); i++)
        for (size_t j = i + 1; j < order.size(); j++)
            if (count[order[j]] > count[order[i]]) std::swap(order[i], order[j]);
    int nl = threads > 1 ? threads : 1;
    for (const Entry& x : r->e)
        if (x.type == E_ALLREDUCE) { nl = 1; break; }       // collectives: one issuing thread, the recorded order on every rank
    if (nl > (int)streams.size()) nl = (int)streams.size() ? (int)streams.size() : 1;
    if (nl > 8) nl = 8;
    r->nlanes = nl;
    std::vector<int> lane_of(streams.size(), nl - 1);
    for (int i = 0; i < (int)order.size(); i++) lane_of[order[i]] = i < nl ? i : nl - 1;
    r->lane_list.assign(nl, {});
    for (int i = 0; i < (int)r->e.size(); i++) {
        size_t k = 0;
        while (streams[k] != r->e[i].stream) k++;
        r->e[i].lane = lane_of[k];
        r->lane_list[lane_of[k]].push_back(i);
    }
    for (int i = (int)r->e.size() - 1; i >= 0; i--) {
        Entry& x = r->e[i];
        if (x.type != E_RECORD) continue;
        if (hipEventCreateWithFlags(&x.own, hipEventDisableTiming) != hipSuccess) return GCC_ERR_LAUNCH;
        x.last = true;
        for (size_t j = i + 1; j < r->e.size(); j++)
            if (r->e[j].type == E_RECORD && r->e[j].ev == x.ev) { x.last = false; break; }
    }
    for (Entry& x : r->e)
        if (x.type == E_WAIT && x.dep >= 0) x.own = r->e[x.dep].own;
    r->issued = new std::atomic<int>[r->e.size() ? r->e.size() : 1];
    for (size_t i = 0; i < r->e.size(); i++) r->issued[i].store(0);
    if (nl > 1) {
        int device = 0;
        (void)hipGetDevice(&device);
        for (int l = 1; l < nl; l++) r->workers.emplace_back(worker_loop, r, l, device);
    }
    return GCC_OK;
}

// marks the next kernel launch the calling thread records (no-op when it is not recording); tag > 0
extern "C" int gcc_replay_tag_next(int tag) {
    if (t_rec) t_rec->pending_tag = tag;
    return GCC_OK;
}

// overwrite argument `arg_index` of every recorded launch that carries `tag`; returns how many were patched (< 0: error)
extern "C" int gcc_replay_patch(gcc_replay_t* r, int tag, int arg_index, const void* value, size_t bytes) {
    if (!r || !r->closed || tag <= 0 || arg_index < 0 || !value) return GCC_ERR_BAD_ARG;
    int n = 0;
    for (const Entry& x : r->e) {
        if (x.type != E_KERNEL || x.tag != tag) continue;
        if ((unsigned)arg_index >= x.nargs || r->arg_size[x.first_arg + arg_index] != bytes) return GCC_ERR_BAD_ARG;
        memcpy(r->arg_ptr[x.first_arg + arg_index], value, bytes);
        n++;
    }
    return n;
}

extern "C" int gcc_replay_run(gcc_replay_t* r) {
    GCC_ENTER();
    if (!r || !r->closed) return GCC_ERR_BAD_ARG;
    if (r->nlanes <= 1) {
        static const bool timing = getenv("GCC_REPLAY_TIMING") != nullptr;     // host time per entry kind, to stderr
        if (timing) {
            double t[6] = {0, 0, 0, 0, 0, 0}; long n[6] = {0, 0, 0, 0, 0, 0};
            for (const Entry& x : r->e) {
                timespec a, b;
                clock_gettime(CLOCK_MONOTONIC, &a);
                if (issue(r, x)) return GCC_ERR_LAUNCH;
                clock_gettime(CLOCK_MONOTONIC, &b);
                t[x.type] += (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; n[x.type]++;
            }
            fprintf(stderr, "[gcc_replay] kernels %ld x %.2f us, memsets %ld x %.2f, copies %ld x %.2f, records %ld x %.2f, waits %ld x %.2f\n",
                    n[0], n[0] ? t[0] / n[0] : 0., n[1], n[1] ? t[1] / n[1] : 0., n[2], n[2] ? t[2] / n[2] : 0., n[3], n[3] ? t[3] / n[3] : 0.,
                    n[4], n[4] ? t[4] / n[4] : 0.);
            return GCC_OK;
        }
        for (const Entry& x : r->e)
            if (issue(r, x)) return GCC_ERR_LAUNCH;
        return GCC_OK;
    }
    int gen;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->failed.store(0);
        r->generation++;
        gen = (int)(r->generation & 0x3fffffff) + 1;
        r->running = r->nlanes - 1;
    }
    r->cv_go.notify_all();
    run_lane(r, 0, gen);
    {
        std::unique_lock<std::mutex> lk(r->mu);
        r->cv_done.wait(lk, [&] { return r->running == 0; });
    }
    return r->failed.load() ? GCC_ERR_LAUNCH : GCC_OK;
}

// what: 0 entries, 1 kernel launches, 2 HIP streams, 3 issuing threads, 4 bytes of argument values
extern "C" long long gcc_replay_info(const gcc_replay_t* r, int what) {
    if (!r) return GCC_ERR_BAD_ARG;
    switch (what) {
    case 0: return (long long)r->e.size();
    case 1: { long long n = 0; for (const Entry& x : r->e) n += x.type == E_KERNEL; return n; }
    case 2: { std::vector<hipStream_t> s; for (const Entry& x : r->e) { bool f = false; for (auto q : s) f |= q == x.stream; if (!f) s.push_back(x.stream); } return (long long)s.size(); }
    case 3: return r->nlanes;
    case 4: return r->closed ? (long long)(r->arg_off.empty() ? 0 : r->arg_off.back() + r->arg_size.back()) : (long long)r->blob.size();
    }
    return GCC_ERR_BAD_ARG;
}

extern "C" int gcc_replay_destroy(gcc_replay_t* r) {
    if (!r) return GCC_ERR_BAD_ARG;
    if (t_rec == r) t_rec = nullptr;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->quit = true;
    }
    r->cv_go.notify_all();
    for (std::thread& t : r->workers) t.join();
    for (Entry& x : r->e)
        if (x.type == E_RECORD && x.own) (void)hipEventDestroy(x.own);
    free(r->args);
    delete[] r->issued;
    delete r;
    return GCC_OK;
}
