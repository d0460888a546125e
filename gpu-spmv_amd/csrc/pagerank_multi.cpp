// pagerank_multi.cpp — row-sharded multi-GPU PageRank in ONE process, behind the reference's API.
//
// The reference has no multi-GPU code; its host loop is src/pagerank.cu:50-153 and its
// public entry point include/spmv/pagerank.h:29-43.  pagerank_multi_gpu() keeps that entry point's
// contract (same config, same result struct, result released by pagerank_free) and shards the CSR row
// range over `num_gpus` devices:
//
//   * boundaries by binary search on row_ptrs for equal nnz (a power-law graph does not leave one device
//     with most of the entries);
//   * device p owns rows [bounds[p], bounds[p + 1]) and a full-length rank vector in the padded layout of
//     csrc/pagerank.hip's shard engine (slice stride = longest block + 4 floats; the 16-byte tail of every
//     slice carries that device's two partial sums as doubles);
//   * per iteration, per device: the fused step kernels (direct vector-CSR or the LDS-tiled engine, the
//     same pr_step the single-GPU loop runs), then ONE in-place ncclAllGather of `stride` floats per device
//     over xGMI (RCCL, single-process ncclCommInitAll, one stream per device), then pr_commit_gathered folds
//     the P partial pairs in device order — identical state on every device, no separate all-reduce;
//   * SPMV_MULTI_GPU=blocks=C, C > 1: the overlapped exchange — the vector is numbered chunk-major (block c =
//     piece c of every shard), C all-gathers per step run on a side stream and the tiled engine's phase 1 of
//     the NEXT step follows them block by block (pr_expand);
//   * one host thread per device drives its stream (issuing for eight devices from one thread would take about as
//     long as the step itself); each runs one step ahead of the convergence check (device-side `done` flag,
//     pinned mirror), as the single-GPU loop does; a failure on any device releases the others
//     (ncclCommAbort) and surfaces as an empty result, not as a hang.
//
// librccl is loaded lazily (dlopen) so that the library itself has no hard dependency on it; with
// num_gpus == 1 the exchange degenerates to nothing and RCCL is not touched unless
// SPMV_MULTI_GPU=force_rccl is set (the single-GPU test box exercises the collective path that way).
// SPMV_MULTI_GPU=share_devices lets more shards than devices run (shard p on device p % available) with the
// slices exchanged by event-ordered device-to-device copies instead of RCCL — RCCL cannot put two ranks on one
// device; this is how the P > 1 partition, layout and commit are tested on a one-GPU box.
#include "internal.h"
#include "pagerank_engine.h"
#include "tiled.h"
#include "spmv/pagerank.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace spmv {

namespace {

constexpr int kTail = 4;        // floats appended to every slice: two doubles of partial sums

struct Rccl {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(lib, "ncclCommAbort"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(lib, "ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(lib, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        api.ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd;
    });
    return api;
}

// host rendezvous of the per-shard threads (copy exchange only); a failing thread releases everybody
class Rendezvous {
public:
    explicit Rendezvous(int parties) : parties_(parties) {}
    // returns false when any party has failed (then or earlier)
    bool arrive(bool fine) {
        std::unique_lock<std::mutex> lock(m_);
        if (!fine) failed_ = true;
        if (failed_) {
            cv_.notify_all();
            return false;
        }
        const long generation = generation_;
        if (++waiting_ == parties_) {
            waiting_ = 0;
            ++generation_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return generation_ != generation || failed_; });
        }
        return !failed_;
    }
    void fail() {
        std::lock_guard<std::mutex> lock(m_);
        failed_ = true;
        cv_.notify_all();
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    const int parties_;
    int waiting_ = 0;
    long generation_ = 0;
    bool failed_ = false;
};

// what one device holds
struct DeviceShard {
    int device = 0;
    hipStream_t stream = nullptr;
    int* d_row_ptrs = nullptr;
    int* d_cols = nullptr;
    float* d_vals = nullptr;
    unsigned char* d_mask = nullptr;
    float* r[2] = {nullptr, nullptr};
    CSRMatrix header;               // wraps the three arrays (side-table key for the tiled plan)
    detail::PrShard shard;
    ncclComm_t comm = nullptr;
    // The communicator is used by this shard's host thread (collectives, under comm_lock) and, on a failure elsewhere,
    // by the failing thread (ncclCommAbort).  The abort never WAITS for comm_lock: a peer blocked on the host inside
    // its collective (connection setup of the first one, a proxy wait) holds that lock until the very abort releases
    // it.  `comm_aborted` is exchanged atomically, so each communicator is aborted exactly once, and the
    // owner looks at it (under its lock) before it starts a collective: an aborted communicator is never used again.
    std::unique_ptr<std::mutex> comm_lock{new std::mutex};
    std::unique_ptr<std::atomic<bool>> comm_aborted{new std::atomic<bool>(false)};      // ncclCommAbort already released it
    bool have_header = false;
    hipStream_t side_stream = nullptr;      // overlapped exchange: the collectives / copies run here
    std::vector<hipEvent_t> block_done;     // ... and block c of the new vector is complete
    hipEvent_t stepped = nullptr;   // this shard's new slice is complete ...
    hipEvent_t gathered = nullptr;  // ... copy exchange: and this shard has finished reading its peers' slices
};

void release(std::vector<DeviceShard>& shards, const Rccl* api) {
    for (DeviceShard& d : shards) {
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (d.comm && api && !d.comm_aborted->load()) (void)api->CommDestroy(d.comm);
        if (d.have_header) detail::aux_drop(d.header.d_row_ptrs);          // the shard's tiled plan, if any
        for (void* p : {static_cast<void*>(d.d_row_ptrs), static_cast<void*>(d.d_cols), static_cast<void*>(d.d_vals),
                        static_cast<void*>(d.d_mask), static_cast<void*>(d.r[0]), static_cast<void*>(d.r[1]),
                        static_cast<void*>(d.shard.d_state), static_cast<void*>(d.shard.d_block_partials)}) {
            if (p) (void)hipFree(p);
        }
        if (d.side_stream) {
            (void)hipStreamSynchronize(d.side_stream);
            (void)hipStreamDestroy(d.side_stream);
        }
        if (d.stream) (void)hipStreamDestroy(d.stream);
        for (hipEvent_t e : d.block_done) if (e) (void)hipEventDestroy(e);
        if (d.stepped) (void)hipEventDestroy(d.stepped);
        if (d.gathered) (void)hipEventDestroy(d.gathered);
    }
    shards.clear();
}

} // namespace

std::vector<int> pagerank_shard_bounds(const int* row_ptrs, int num_rows, int num_shards) {
    std::vector<int> bounds(static_cast<size_t>(num_shards) + 1, 0);
    bounds[num_shards] = num_rows;
    const long long nnz = row_ptrs[num_rows];
    for (int p = 1; p < num_shards; ++p) {
        const long long target = (nnz * p + num_shards / 2) / num_shards;
        int at = static_cast<int>(std::lower_bound(row_ptrs, row_ptrs + num_rows + 1, target) - row_ptrs);
        at = std::min(at, num_rows);
        // the search lands on the first boundary at or past the target; the one before may be nearer
        if (at > 0 && target - row_ptrs[at - 1] < row_ptrs[at] - target) --at;
        bounds[p] = std::max(at, bounds[p - 1]);
    }
    return bounds;
}

PageRankResult pagerank_multi_gpu(const CSRMatrix* adj, const PageRankConfig* config, int num_gpus) {
    PageRankResult result;
    if (!adj) return result;
    const PageRankConfig fallback;
    if (!config) config = &fallback;
    const int n = adj->num_rows;
    if (n <= 0 || num_gpus < 1 || !adj->row_ptrs || (adj->nnz > 0 && (!adj->col_indices || !adj->values))) {
        // the shards are cut from the HOST arrays (csr_from_dense / csr_deserialize / csr_from_gpu provide them)
        return n <= 0 || num_gpus < 1 ? pagerank(adj, config) : result;
    }
    int available = 0;
    // SPMV_MULTI_GPU="blocks=C,share_devices,force_rccl" — blocks: the overlapped exchange (below); share_devices: more shards
    // than devices (exchange by device copies: how P > 1 is tested on a one-GPU box); force_rccl: keep the collective in the
    // loop with ONE shard (the RCCL call itself on a one-GPU box)
    const char* options = std::getenv("SPMV_MULTI_GPU");
    const bool share_devices = detail::list_option(options, "share_devices");
    if (hipGetDeviceCount(&available) != hipSuccess || available < 1 || (available < num_gpus && !share_devices)) {
        (void)hipGetLastError();
        return result;
    }
    const bool by_copies = share_devices && available < num_gpus;     // several shards per device: no RCCL
    int previous_device = 0;
    (void)hipGetDevice(&previous_device);

    const int P = num_gpus;
    const bool force_rccl = detail::list_option(options, "force_rccl");
    const bool exchange = P > 1 || force_rccl;
    const Rccl* api = exchange && !by_copies ? &rccl() : nullptr;
    if (api && !api->ok) {
        std::fprintf(stderr, "spmv: librccl not available, pagerank_multi_gpu cannot exchange rank slices\n");
        return result;
    }

    // ---- partition: equal nnz; padded layout of the rank vector
    const std::vector<int> bounds = pagerank_shard_bounds(adj->row_ptrs, n, P);
    int longest = 0;
    for (int p = 0; p < P; ++p) longest = std::max(longest, bounds[p + 1] - bounds[p]);
    if (longest % 2) ++longest;                                  // 8-byte aligned tails
    // The vector is a sequence of `blocks` blocks, block c = piece c of every shard back to back (see RowMap in
    // pagerank_engine.h and Layout in pagerank_dist.py).  One block (default): every shard's slice is contiguous,
    // one all-gather per step.  SPMV_MULTI_GPU=blocks=C, C > 1: the overlapped exchange — C all-gathers on a side
    // stream, the next step's phase 1 for the columns of block c runs while block c + 1 is on the links.
    int blocks = 1;
    {
        long long wanted = 1;
        if (detail::list_option(options, "blocks", &wanted)) blocks = static_cast<int>(std::max(1LL, std::min(16LL, wanted)));
    }
    if (!exchange) blocks = 1;
    const int tail = exchange ? kTail : 0;
    long long piece = longest + tail;
    if (blocks > 1) {
        const long long align = longest >= (1 << 18) ? 32768 : 4;   // block boundaries = strip boundaries of the tiled engine
        const long long per_block = (static_cast<long long>(longest) + tail + blocks - 1) / blocks;
        piece = (per_block + align - 1) / align * align;
    }
    const long long block = piece * P;
    const long long padded = block * blocks;
    if (padded > 0x7fffffffLL) return result;
    auto place = [&](int owner, long long i) {                   // (shard, row inside it) -> index in the padded vector
        return static_cast<int>(blocks == 1 ? owner * piece + i : (i / piece) * block + owner * piece + i % piece);
    };
    auto position = [&](int node) {                              // node -> index in the padded vector
        const int owner = static_cast<int>(std::upper_bound(bounds.begin() + 1, bounds.end() - 1, node) - (bounds.begin() + 1));
        return place(owner, node - bounds[owner]);
    };
    const long long last_block = (blocks - 1) * block;           // the tails sit at the end of every shard's last piece

    // dangling columns exactly as the reference's host scan (src/pagerank.cu:20-48), in padded positions
    std::vector<unsigned char> mask(static_cast<size_t>(padded), 0);
    unsigned long long num_dangling = 0;
    {
        std::vector<float> sums(static_cast<size_t>(adj->num_cols), 0.0f);
        for (int r = 0; r < n; ++r) {
            for (int j = adj->row_ptrs[r]; j < adj->row_ptrs[r + 1]; ++j) {
                const int c = adj->col_indices[j];
                if (c >= 0 && c < adj->num_cols) sums[c] += adj->values[j];
            }
        }
        for (int c = 0; c < std::min(n, adj->num_cols); ++c) {
            if (sums[c] == 0.0f) {
                mask[position(c)] = 1;
                ++num_dangling;
            }
        }
    }
    const float start = 1.0f / n;
    detail::PrState first_state{};
    for (unsigned long long k = 0; k < num_dangling; ++k) first_state.dangling_sum += start;
    std::vector<float> start_vector(static_cast<size_t>(padded), 0.0f);
    for (int p = 0; p < P; ++p) {
        for (int i = 0; i < bounds[p + 1] - bounds[p]; ++i) start_vector[place(p, i)] = start;
    }

    // ---- upload the shards
    std::vector<DeviceShard> shards(P);
    bool ok = true;
    std::vector<int> local_cols;
    for (int p = 0; p < P && ok; ++p) {
        DeviceShard& d = shards[p];
        d.device = p % available;
        const int rows = bounds[p + 1] - bounds[p];
        const int first = adj->row_ptrs[bounds[p]];
        const int local_nnz = adj->row_ptrs[bounds[p + 1]] - first;
        std::vector<int> local_ptrs(static_cast<size_t>(rows) + 1);
        for (int r = 0; r <= rows; ++r) local_ptrs[r] = adj->row_ptrs[bounds[p] + r] - first;
        local_cols.resize(static_cast<size_t>(std::max(local_nnz, 1)));
        for (int j = 0; j < local_nnz; ++j) {
            const int c = adj->col_indices[first + j];
            local_cols[j] = c >= 0 && c < n ? position(c) : 0;   // node ids -> positions in the padded vector
        }
        ok = hipSetDevice(d.device) == hipSuccess
          && hipStreamCreate(&d.stream) == hipSuccess
          && (blocks == 1 || hipStreamCreate(&d.side_stream) == hipSuccess)
          && hipEventCreateWithFlags(&d.stepped, hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&d.gathered, hipEventDisableTiming) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.d_row_ptrs), (static_cast<size_t>(rows) + 1) * sizeof(int)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.d_cols), static_cast<size_t>(std::max(local_nnz, 1)) * sizeof(int)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.d_vals), static_cast<size_t>(std::max(local_nnz, 1)) * sizeof(float)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.d_mask), static_cast<size_t>(padded)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.r[0]), static_cast<size_t>(padded) * sizeof(float)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&d.r[1]), static_cast<size_t>(padded) * sizeof(float)) == hipSuccess
          && hipMemcpy(d.d_row_ptrs, local_ptrs.data(), local_ptrs.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess
          && hipMemcpy(d.d_cols, local_cols.data(), static_cast<size_t>(local_nnz) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess
          && hipMemcpy(d.d_vals, adj->values + first, static_cast<size_t>(local_nnz) * sizeof(float), hipMemcpyHostToDevice) == hipSuccess
          && hipMemcpy(d.d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice) == hipSuccess
          && hipMemcpy(d.r[0], start_vector.data(), start_vector.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess
          && hipMemcpy(d.r[1], start_vector.data(), start_vector.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) break;
        std::memset(&d.header, 0, sizeof(d.header));
        d.header.num_rows = rows;
        d.header.num_cols = static_cast<int>(padded);
        d.header.nnz = local_nnz;
        d.header.d_row_ptrs = d.d_row_ptrs;
        d.header.d_col_indices = d.d_cols;
        d.header.d_values = d.d_vals;
        d.have_header = true;
        detail::PrShard& sh = d.shard;
        sh.local_rows = rows;
        sh.map.base = static_cast<int>(p * piece);
        if (blocks > 1) {
            sh.map.piece = static_cast<int>(piece);
            sh.map.block = static_cast<int>(block);
        }
        d.block_done.assign(blocks > 1 ? blocks : 0, nullptr);
        for (hipEvent_t& e : d.block_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        sh.n_global = n;
        sh.nnz = local_nnz;
        sh.d_row_ptrs = d.d_row_ptrs;
        sh.d_cols = d.d_cols;
        sh.d_vals = d.d_vals;
        sh.d_dangling = d.d_mask;
        const int pairs = detail::pr_shard_prepare(&sh, detail::tiled_plan_for(&d.header, d.stream));
        ok = hipMalloc(reinterpret_cast<void**>(&sh.d_state), sizeof(detail::PrState)) == hipSuccess
          && hipMalloc(reinterpret_cast<void**>(&sh.d_block_partials), 2 * sizeof(double) * static_cast<size_t>(pairs)) == hipSuccess
          && hipMemcpy(sh.d_state, &first_state, sizeof(first_state), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && exchange && !by_copies) {
        std::vector<int> devices(P);
        std::vector<ncclComm_t> comms(P, nullptr);
        for (int p = 0; p < P; ++p) devices[p] = p;
        ok = api->CommInitAll(comms.data(), P, devices.data()) == ncclSuccess;
        for (int p = 0; p < P; ++p) shards[p].comm = comms[p];
    }

    // ---- the loop: ONE HOST THREAD PER SHARD (a step is a handful of launches per device; issued from one thread
    // for eight devices they take about as long as the step itself).  Every thread enqueues step k + 1 before it
    // looks at the outcome of step k (pinned mirror of its own device's state; the states are identical on all
    // devices, so all threads leave the loop after the same step).  With RCCL nothing but the collective couples
    // the threads; the copy exchange needs two host rendezvous per step (an event must have been recorded before
    // another thread can wait on it).
    detail::PrState last_state{};
    Rendezvous meet(P);
    std::atomic<bool> aborted{false};       // some thread failed: nobody starts another collective
    std::vector<detail::PrState*> pinned(P, nullptr);
    std::vector<std::array<hipEvent_t, 2>> seen(P, std::array<hipEvent_t, 2>{nullptr, nullptr});
    for (int p = 0; p < P && ok; ++p) {
        ok = hipSetDevice(shards[p].device) == hipSuccess
          && hipHostMalloc(reinterpret_cast<void**>(&pinned[p]), 2 * sizeof(detail::PrState)) == hipSuccess
          && hipEventCreateWithFlags(&seen[p][0], hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&seen[p][1], hipEventDisableTiming) == hipSuccess;
    }
    auto drive = [&](int p) {
        DeviceShard& d = shards[p];
        bool fine = hipSetDevice(d.device) == hipSuccess;
        for (int iter = 0; fine && !aborted.load() && iter < config->max_iterations; ++iter) {
            const float* r_old = d.r[iter & 1];
            float* r_new = d.r[(iter + 1) & 1];
            if (by_copies && iter > 0) {
                // This step writes r_new = the vector the peers copied this shard's slice FROM during the exchange
                // of step iter - 2.  Their `gathered` events (last recorded after the exchange of step iter - 1,
                // later in the same streams; the rendezvous that ended that step made sure they are recorded).
                for (int q = 0; q < P && fine; ++q) {
                    if (q != p) fine = hipStreamWaitEvent(d.stream, shards[q].gathered, 0) == hipSuccess;
                }
            }
            fine = fine && detail::pr_step(d.shard, r_old, r_new, config->damping_factor, detail::PushTargets{}, d.stream) == hipSuccess;
            if (fine && exchange) {      // the two partial sums go into the tail of this device's own slice
                double* tail_slot = reinterpret_cast<double*>(r_new + last_block + (p + 1) * piece - kTail);
                fine = detail::pr_reduce(d.shard, tail_slot, d.stream) == hipSuccess;
            } else if (fine) {
                fine = detail::pr_reduce_commit(d.shard, config->tolerance, d.stream) == hipSuccess;
            }
            if (fine && exchange) {
                // The exchange: block by block (one block unless SPMV_MULTI_GPU=blocks=C), on the side stream when
                // there are several, so that this stream can multiply block c while block c + 1 travels.
                const bool side = blocks > 1;
                hipStream_t xs = side ? d.side_stream : d.stream;
                if (side || by_copies) fine = hipEventRecord(d.stepped, d.stream) == hipSuccess;
                if (fine && side) fine = hipStreamWaitEvent(xs, d.stepped, 0) == hipSuccess;
                if (by_copies && !meet.arrive(fine)) { fine = false; break; }        // every `stepped` is recorded
                for (int c = 0; c < blocks && fine; ++c) {
                    float* mine = r_new + c * block;
                    if (by_copies) {      // pull piece c of every peer (tail included in the last one)
                        for (int q = 0; q < P && fine; ++q) {
                            if (q == p) continue;
                            if (c == 0) fine = hipStreamWaitEvent(xs, shards[q].stepped, 0) == hipSuccess;
                            fine = fine && hipMemcpyAsync(mine + q * piece, shards[q].r[(iter + 1) & 1] + c * block + q * piece,
                                                          static_cast<size_t>(piece) * sizeof(float), hipMemcpyDeviceToDevice, xs) == hipSuccess;
                        }
                    } else {
                        std::lock_guard<std::mutex> mine_only(*d.comm_lock);
                        fine = !aborted.load() && !d.comm_aborted->load()
                            && api->AllGather(mine + p * piece, mine, static_cast<size_t>(piece), ncclFloat, d.comm, xs) == ncclSuccess;
                    }
                    if (fine && side) {
                        fine = hipEventRecord(d.block_done[c], xs) == hipSuccess
                            && hipStreamWaitEvent(d.stream, d.block_done[c], 0) == hipSuccess;
                        // head start on the next step: the columns of the blocks that have arrived
                        if (fine && c + 1 < blocks) fine = detail::pr_expand(d.shard, r_new, static_cast<long long>(c + 1) * block, d.stream) == hipSuccess;
                    }
                }
                if (by_copies) {
                    fine = fine && hipEventRecord(d.gathered, xs) == hipSuccess;
                    if (!meet.arrive(fine)) { fine = false; break; }                 // every `gathered` is recorded
                }
                fine = fine && detail::pr_commit_gathered(d.shard, r_new + last_block, P, piece, piece - kTail, config->tolerance,
                                                          d.stream) == hipSuccess;
            }
            fine = fine
                && hipMemcpyAsync(&pinned[p][iter & 1], d.shard.d_state, sizeof(detail::PrState), hipMemcpyDeviceToHost, d.stream) == hipSuccess
                && hipEventRecord(seen[p][iter & 1], d.stream) == hipSuccess;
            if (fine && iter >= 1) {
                fine = hipEventSynchronize(seen[p][(iter - 1) & 1]) == hipSuccess;
                if (fine && pinned[p][(iter - 1) & 1].done) break;
            }
        }
        if (!fine) {
            // Let nobody wait for this thread: release the rendezvous, and abort the collectives the peers may be
            // blocked in on the device (a rank that never joins would otherwise hang them).
            meet.fail();
            aborted.store(true);
            if (api && api->CommAbort) {
                for (DeviceShard& other : shards) {
                    if (!other.comm) continue;
                    // Owner not inside a collective: abort under its lock, so that it cannot start one meanwhile.  Owner
                    // inside one (the lock is taken): that call may be exactly what the abort has to release — ncclCommAbort
                    // is the one call meant to be issued against a communicator in use —, so do not wait for the lock.
                    std::unique_lock<std::mutex> idle(*other.comm_lock, std::try_to_lock);
                    if (!other.comm_aborted->exchange(true)) (void)api->CommAbort(other.comm);     // each communicator exactly once
                }
            }
        }
        fine = hipStreamSynchronize(d.stream) == hipSuccess && fine;
        if (d.side_stream) fine = hipStreamSynchronize(d.side_stream) == hipSuccess && fine;
        return fine;
    };
    if (ok) {
        std::vector<char> outcome(P, 0);
        if (P == 1) {
            outcome[0] = drive(0);
        } else {
            std::vector<std::thread> threads;
            for (int p = 0; p < P; ++p) threads.emplace_back([&, p] { outcome[p] = drive(p); });
            for (std::thread& t : threads) t.join();
        }
        for (int p = 0; p < P; ++p) ok = ok && outcome[p];
        // (a failing thread has aborted every communicator and marked it: release() destroys only the others)
    }
    if (ok) {
        ok = hipSetDevice(shards[0].device) == hipSuccess
          && hipMemcpy(&last_state, shards[0].shard.d_state, sizeof(last_state), hipMemcpyDeviceToHost) == hipSuccess;
    }

    // ---- result: the last written vector, slices compacted, r /= sum(r) accumulated in double
    if (ok) {
        result.iterations = last_state.iterations;
        result.final_residual = last_state.final_residual;
        result.converged = last_state.converged != 0;
        result.ranks = new float[n];
        const float* last = shards[0].r[last_state.iterations & 1];
        for (int p = 0; p < P && ok; ++p) {               // every shard's rows, piece by piece
            const long long rows = bounds[p + 1] - bounds[p];
            for (long long done = 0; done < rows && ok; done += piece) {
                const long long count = std::min<long long>(piece, rows - done);
                ok = hipMemcpy(result.ranks + bounds[p] + done, last + place(p, done), static_cast<size_t>(count) * sizeof(float),
                               hipMemcpyDeviceToHost) == hipSuccess;
            }
        }
        if (ok) {
            double total = 0.0;
            for (int i = 0; i < n; ++i) total += result.ranks[i];
            const float divisor = static_cast<float>(total);
            if (divisor > 0.0f) for (int i = 0; i < n; ++i) result.ranks[i] /= divisor;
        } else {
            delete[] result.ranks;
            result = PageRankResult();
        }
    }
    for (int p = 0; p < P; ++p) {
        if (pinned[p]) (void)hipHostFree(pinned[p]);
        for (hipEvent_t e : seen[p]) if (e) (void)hipEventDestroy(e);
    }
    release(shards, api);
    (void)hipSetDevice(previous_device);
    (void)hipGetLastError();
    return result;
}

} // namespace spmv
