// Device-resident min-set-cover gather (gather_build.hip: the index; gather.hip: the rounds).  Raw device pointers; owns its index and counters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "qindex.hpp"

namespace smg {

// slots of the 16 x u64 device state block
enum GatherSlot {
    GS_KEY = 0,      // packed winner of the round in flight: (count << 32) | (0xffffffff & ~global index)
    GS_DONE = 1,     // 1 once a stop rule fired
    GS_ROUNDS = 2,   // results recorded so far
    GS_QLEN = 3,     // hashes of the query still uncovered
    GS_ACC = 4,      // |I| of the round in flight
    GS_PENDING = 5,  // a round was applied and is not recorded yet
    GS_THR = 6,      // minimum overlap in hashes (ceil(threshold_bp / scaled))
    GS_MAXR = 7,     // maximum number of results
    GS_TICKET = 8,   // workgroups of the running pick / top-K kernel that have delivered their partial result
    GS_NEEDX = 9,    // replay: the best candidate fell below what a rank kept back -> the next exchange decides
    GS_WSLOT = 10,   // replay: candidate slot of the round's winner
    GS_BOUND = 11,   // replay: largest key any rank kept back at the last exchange (0: nothing kept back)
    GS_ERR = 12,     // persistent loop: gave up.  10: the grid was not resident as a whole within the gate's time (nothing touched);
                     // 11 / 12 / 13: a workgroup / a rank / the rank's own workgroup 0 did not show up in a sweep, 14: another rank
                     // called the run off -- all between two rounds, the state is that of GS_ROUNDS whole rounds;
                     // 2 / 3 / 4: a winner's row did not arrive or does not fit the exchange (the run is void)
    GS_SLOTS = 16
};

struct GatherDev {
    // borrowed (must outlive the object)
    const uint64_t* Q = nullptr;        // sorted unique query hashes
    uint64_t nq = 0;
    const uint64_t* hashes = nullptr;   // CSR database shard
    const uint64_t* offsets = nullptr;
    uint64_t ndb = 0, index_base = 0;
    hipStream_t stream = nullptr;       // stream of the build: the owned buffers are arena blocks (arena.hpp) released on it
    // owned
    uint64_t* q_padded = nullptr;       // copy of Q followed by 4 copies of its last element (lookups read 4 entries at once)
    uint32_t* q_table = nullptr;        // [q_buckets + 1] first-level table over Q: bucket b = x >> q_shift
    QRec* q_rec = nullptr;              // [q_buckets] bucket records (position, count, first three hashes inline): lookups read these
    uint32_t q_shift = 0, q_buckets = 1;
    uint64_t q_max = 0;                 // Q[nq - 1]
    uint8_t* alive = nullptr;           // [nq] 1 while the query hash is uncovered
    uint64_t* post_off = nullptr;       // [nq + 1] postings of query hash j: post_rows[post_off[j] .. post_off[j+1])
    uint32_t* post_rows = nullptr;      // local row ids
    uint32_t* qpos = nullptr;           // [database elements] position in Q of every element of the shard (NONE32: not in Q):
                                        // a round applied from the local CSR skips the lookup (two dependent loads)
    uint32_t* block_pre = nullptr;      // [nq][block_B + 1] absolute start of row block b's run inside posting list j (last: the list's end)
                                        // (staged range build with block-ordered lists only): the persistent loop's workgroups
                                        // read only their rows' part of a list
    uint32_t block_B = 0, block_rows = 0;
    unsigned long long* loop_xchg = nullptr;   // persistent loop: [2][workgroups][4] record granules
    uint32_t loop_wgs = 0;
    size_t loop_words = 0;
    bool counters_touched = false;      // a caller overwrote counters (smgpu_counter_set): the fused loops keep to saturating steps
    uint64_t npairs = 0;
    uint64_t longest_row = 0;           // hashes in the shard's longest row (sizes the candidate records)
    unsigned long long* counters = nullptr;   // [ndb] |row_d ∩ uncovered query|
    unsigned long long* state = nullptr;      // [GS_SLOTS]
    unsigned long long* partials = nullptr;   // [GATHER_PICK_BLOCKS]
    uint64_t* out_idx = nullptr;        // [out_cap] global index of the round's winner
    uint64_t* out_isect = nullptr;      // [out_cap] |I| of the round
    uint64_t out_cap = 0;
    // candidate replay (see gather_topk_export): allocated on first use
    unsigned long long* topk_sel = nullptr;       // [GATHER_TOPK_MAX + 1] best local keys, descending; the last one is kept back
    unsigned long long* topk_partials = nullptr;  // [GATHER_PICK_BLOCKS][GATHER_TOPK_MAX + 1]
    uint64_t* cmask = nullptr;          // [nq] bit c set: candidate c of the current exchange holds query hash j
    unsigned long long* cand_count = nullptr;     // [64] |candidate row ∩ uncovered query|, kept exact by apply
    unsigned long long* cand_key = nullptr;       // [64] key the candidate was exported with (its global index)
    uint32_t* cand_len = nullptr;       // [64] row lengths of the loaded candidates (to take their bits out again)
    uint32_t* cand_qpos = nullptr;      // [64][cand_qstride] query positions of the candidates' hashes
    uint64_t cand_qstride = 0;
    const uint64_t* cands = nullptr;    // borrowed: the gathered candidate records of the current exchange
    uint64_t cand_stride = 0;           // u64 words per record: [key, bound, len, hashes...]
    uint32_t n_cand = 0;
    uint64_t* own_cands = nullptr;      // single-GPU loop: this shard's own export buffer
    uint64_t own_cands_words = 0;
    // what the build cost (smgpu_gather_build_stats): host wall clock, kernel span between two events on the build's stream,
    // driver allocator calls made through the arena during the build, host synchronisations
    unsigned long long* pinned = nullptr;  // 32 x u64 of pinned host memory: scalar read-backs land here
    hipEvent_t ev_build0 = nullptr, ev_build1 = nullptr;
    uint64_t loop_fallbacks = 0;        // times the resident loop gave up (grid not resident as a whole) and other rounds took over
    double loop_gpu_ms = 0.0;
    uint64_t loop_host_ns = 0;
    uint64_t build_host_ns = 0, build_driver_ns = 0, build_driver_allocs = 0, build_syncs = 0, build_sync_wait_ns = 0;
    hipGraphExec_t loop_graph = nullptr;   // GATHER_GRAPH_ROUNDS rounds of pick + apply, captured once (hosts that launch slowly)
    hipStream_t loop_stream = nullptr;     // the graph's own stream (the legacy default stream cannot capture)
};

constexpr unsigned GATHER_GRAPH_ROUNDS = 64;

constexpr unsigned GATHER_PICK_BLOCKS = 256;
constexpr unsigned GATHER_TOPK_MAX = 16;     // candidates a rank can export per exchange
constexpr unsigned GATHER_CAND_MAX = 64;     // candidates of all ranks together (one bit each in cmask)
constexpr unsigned GATHER_CAND_HEAD = 3;     // record header words: key, bound, len

// Build the inverted index and the initial counters.  Synchronises the stream twice (sizes of the table / of the
// postings must reach the host to size buffers) and returns with the last kernels still in flight on `stream`.
// Every buffer comes from the arena: a rebuild of the same shape makes no driver call.
hipError_t gather_build(GatherDev& g, hipStream_t stream);
// milliseconds between the first and the last kernel of the build (waits for the build to finish)
hipError_t gather_build_kernel_ms(GatherDev& g, float* ms);
void gather_destroy(GatherDev& g);
// Arm the loop: thresholds, result capacity (reallocated if too small), state reset except alive/counters.
hipError_t gather_begin(GatherDev& g, uint64_t thr_hashes, uint64_t max_rounds, hipStream_t stream);
// Record the previous round (if one is pending), then the local best -> state[GS_KEY] and *d_key_out (may be null).
// check_stop != 0: also evaluate the stop rules on that key (single-GPU loop).
hipError_t gather_pick(GatherDev& g, unsigned long long* d_key_out, int check_stop, hipStream_t stream);
// Apply the round: I = row ∩ uncovered query; uncovered -= I; counters[d] -= |I ∩ row_d| through the postings.
// The winner of state[GS_KEY] is read from the local CSR.
hipError_t gather_apply(GatherDev& g, hipStream_t stream);
// ---- candidate replay: several rounds per exchange (sharded databases; also the single-GPU loop) ----
// The K best local rows (by the packed key) as records [key, bound, len, hashes...] of `stride` u64 words into
// d_out[K][stride]; `bound` = the best key this shard keeps back (0: none).  stride >= 3 + longest row.
hipError_t gather_topk_export(GatherDev& g, uint64_t* d_out, uint32_t K, uint64_t stride, hipStream_t stream);
// Adopt the gathered records of all shards (n_cand <= 64, same stride; borrowed until the next load): their hashes'
// query positions become bits in cmask, their counters start from the exported keys.
hipError_t gather_cands_load(GatherDev& g, const uint64_t* d_cands, uint32_t n_cand, uint64_t stride, hipStream_t stream);
// `rounds` rounds of: winner among the candidates (valid while it beats every kept-back key; stop rules) + apply to
// the local postings and to the candidates' counters.  No-ops once a stop rule fired or an exchange is needed.
hipError_t gather_replay_rounds(GatherDev& g, unsigned rounds, hipStream_t stream);
// single-GPU loop built from the three calls above (export into an own buffer, no collective)
hipError_t gather_enqueue_replay(GatherDev& g, unsigned exchanges, hipStream_t stream);
// CounterGather.consume for a caller-provided list ([len, hashes...] on device, every hash a member of Q):
// counters[d] -= |list ∩ row_d| (saturating at 0), and the hashes leave the uncovered set.
hipError_t gather_consume_list(GatherDev& g, const uint64_t* d_list, hipStream_t stream);
// The whole armed loop as ONE resident kernel (gather.hip: gather_loop_kernel), when the index was built by the staged
// range builder and the query's bitmap + a workgroup's rows fit LDS; *ran = false (and nothing launched) otherwise.
// The caller synchronises the stream and reads the state block as after any batch of rounds.
hipError_t gather_run_persistent(GatherDev& g, hipStream_t stream, bool* ran);
// Several ranks (one process and database shard each) running the SAME rounds: the local winners meet in host-visible memory
// shared by all ranks (pinned; POSIX shared memory registered with HIP when the ranks are processes), the best of them is
// the round's winner everywhere, and its query positions reach the other ranks through the same memory -- no host
// collective inside the loop.  rec: [2][W][4] granules, rows: [2][W][rowcap] granules (device-visible addresses).
constexpr uint32_t GATHER_PEERS_MAX = 16;    // ranks of one node that can exchange through per-rank device areas
struct GatherShared {
    unsigned long long* rec;
    unsigned long long* rows;
    unsigned long long* const* peers = nullptr;   // non-null: W device-visible addresses, area r = rank r's own device memory
                                                  // ([2][4] record granules + [2][rowcap] row granules); rec / rows unused
    uint32_t W, rank;
    uint64_t rowcap;              // >= the longest row of any rank's shard
    uint32_t run_id;              // the same on every rank, different from the previous run on this memory
};
// the persistent loop on n_wg workgroups (0: one per CU), alone (sh == nullptr) or as rank sh->rank of sh->W
hipError_t gather_launch_loop(GatherDev& g, hipStream_t stream, uint32_t n_wg, const GatherShared* sh, bool* ran);
bool gather_loop_eligible(const GatherDev& g, uint32_t n_wg);
// test support: n_wg workgroups holding lds_bytes of LDS each for `micros` microseconds on `stream` (somebody else's kernel)
hipError_t debug_hold_cus(uint32_t n_wg, uint32_t lds_bytes, uint64_t micros, hipStream_t stream);
hipError_t gather_loop_reserve(GatherDev& g, hipStream_t stream, uint32_t n_wg, uint64_t rowcap);
// Enqueue `rounds` rounds of pick(check) + apply on one GPU (kernels are no-ops once GS_DONE is set).
hipError_t gather_enqueue_rounds(GatherDev& g, unsigned rounds, hipStream_t stream);
// The same rounds as replays of one captured graph of GATHER_GRAPH_ROUNDS rounds (rounded up): one host call per 64
// rounds instead of 128 launches.  Slower than eager launches on a quiet host, faster when the host cannot keep the
// queue ahead of the kernels (profiles/r02_gather_host_variance.txt); the caller decides.
hipError_t gather_enqueue_rounds_graph(GatherDev& g, unsigned rounds, hipStream_t* used);

// overlap[d] = |Q ∩ D_d| (op 0) or overlap[d] -= |Q ∩ D_d| saturating (op 1) by the streaming walks of overlap.hip: the form of
// pair_api.hpp's overlap_vector_launch for large queries over many rows (synchronises the stream twice); hipErrorNotSupported when
// the query's ranges fit neither streaming form
hipError_t overlap_ranges_launch(const uint64_t* Q, uint64_t nq, const uint64_t* hashes, const uint64_t* offsets, uint64_t ndb,
                                 unsigned long long* overlap, int op, hipStream_t stream);
constexpr uint64_t OVERLAP_RANGES_MIN_NQ = 4 * 32768;   // below this the one-wave-per-row kernel's table already sits in L2
constexpr uint64_t OVERLAP_RANGES_MIN_ROWS = 4096;

}  // namespace smg
