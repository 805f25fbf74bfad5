// Device group: one engine per GPU, objects sharded over the ranks, RCCL only to gather the finished audio
// (include/openpbso_amd.h "device group").  Objects never interact (modal_solver.h:100-126), so
// stepping needs no exchange; this file is host-side plumbing over the C ABI of the single engine plus three collectives.
// librccl is loaded at run time (dlopen) when a group of more than one rank is created -- or when a test asks for the
// communicator of a one-rank group (PBSO_GROUP_RCCL_ALWAYS) --: an engine alone does not need it, and this file does not
// need RCCL's headers to build (the dozen declarations it uses are below).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/openpbso_amd.h"
#include "kernels.h"

namespace {

// ---- the part of RCCL's C API this file calls (rccl/rccl.h; the values are NCCL's ABI)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[PBSO_GROUP_ID_BYTES]; } ncclUniqueId;
// (fixed underlying type: the library returns codes this file does not name -- values outside a plain enum's range would be
//  undefined behaviour)
enum ncclResult_t : int { ncclSuccess = 0 };
enum ncclDataType_t : int { ncclFloat = 7 };
enum ncclRedOp_t : int { ncclSum = 0 };
static_assert(sizeof(ncclUniqueId) == 128, "NCCL_UNIQUE_ID_BYTES");

struct Rccl {
    void *lib = nullptr;
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    int version = 0;
    std::string err;
    bool load() {
        if (lib) return ok;                              // (a library that lacks a symbol stays unusable: no second try through null pointers)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
        bool all = true;
        auto sym = [&](auto &fp, const char *n) {
            fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(lib, n));
            if (!fp) { all = false; err = std::string("librccl lacks ") + n; }
        };
        sym(GetUniqueId, "ncclGetUniqueId"); sym(CommInitRank, "ncclCommInitRank"); sym(CommInitAll, "ncclCommInitAll");
        sym(CommDestroy, "ncclCommDestroy"); sym(AllGather, "ncclAllGather"); sym(AllReduce, "ncclAllReduce");
        sym(Send, "ncclSend"); sym(Recv, "ncclRecv"); sym(GroupStart, "ncclGroupStart"); sym(GroupEnd, "ncclGroupEnd");
        sym(GetErrorString, "ncclGetErrorString"); sym(GetVersion, "ncclGetVersion");
        // the declarations above are NCCL 2's ABI (ncclFloat = 7, a 128-byte id, ncclSum = 0: unchanged since 2.0): refuse anything else
        if (all && (GetVersion(&version) != ncclSuccess || version < 2000)) {
            all = false;
            err = "librccl reports version " + std::to_string(version) + ": not the NCCL 2 ABI this file declares";
        }
        ok = all;
        return ok;
    }
};
Rccl g_rccl;

struct Rank {
    int rank = 0, device = 0;
    pbso_engine *eng = nullptr;
    hipStream_t stream = nullptr, coll = nullptr;        // the engine's launch stream; the collective's stream
    ncclComm_t comm = nullptr;
    float *target[2] = {nullptr, nullptr};               // gather targets [rows_total][row] (ALL / ROOT on rank 0) or the rank's own rows
    float *mix[2] = {nullptr, nullptr};                  // [row]
    float *scratch = nullptr;                            // RCCL_ALWAYS, one rank: where the send to itself lands; LOOPBACK: [world][row] mix rows + the sum's partial rows
    size_t target_floats = 0, mix_floats = 0, scratch_floats = 0;
    hipEvent_t ev_step[2] = {nullptr, nullptr}, ev_coll[2] = {nullptr, nullptr};
    int n_local = 0, next_local = 0;
};

}  // namespace

struct pbso_group {
    pbso_engine_desc edesc;
    std::vector<Rank> ranks;                             // the LOCAL ranks
    int world = 0, first = 0;
    std::vector<int> cuts;                               // world + 1 cut points: rank r owns global ids [cuts[r], cuts[r + 1])
    int n_objects = 0, cmax = 0;
    bool planned = false, finalized = false;
    int slot = 0, last_nb = 0, last_mode = 0, last_slot = -1;
    bool stepped = false;
    int transport = PBSO_GROUP_RCCL;
    bool use_rccl = false;                               // a communicator exists: the collectives go through librccl
    bool loopback() const { return transport == PBSO_GROUP_LOOPBACK; }
    std::string err;
    int frames = PBSO_FRAMES_PER_BUFFER;
};

namespace {

int gfail(pbso_group *g, int code, const std::string &m) {
    if (g) g->err = m;
    return code;
}
#define GHIP(g, expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return gfail(g, PBSO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define GNCCL(g, expr)                                                                            \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) return gfail(g, PBSO_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
    } while (0)
// ... between ncclGroupStart and ncclGroupEnd: the group is closed before the error is returned (an open group would swallow
// every later call of this thread)
#define GNCCL_OPEN(g, expr)                                                                       \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            (void)g_rccl.GroupEnd();                                                              \
            return gfail(g, PBSO_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
        }                                                                                         \
    } while (0)
#define GHIP_OPEN(g, expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)g_rccl.GroupEnd();                                                              \
            return gfail(g, PBSO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
        }                                                                                         \
    } while (0)
#define GENG(g, rk, expr)                                                                         \
    do {                                                                                          \
        int _c = (expr);                                                                          \
        if (_c < 0) return gfail(g, _c, std::string("rank ") + std::to_string((rk).rank) + ": " + pbso_last_error((rk).eng)); \
    } while (0)

// loopback all-reduce: [world mix rows | partial rows of the sum | the result row]
float *loop_mix_out(const pbso_group *g, const Rank &rk, size_t row) {
    return rk.scratch + (size_t)(g->world + pbso::mix_objects_groups(g->world)) * row;
}

Rank *local_rank(pbso_group *g, int rank) {
    const int i = rank - g->first;
    return i >= 0 && i < (int)g->ranks.size() ? &g->ranks[i] : nullptr;
}

int ensure_buffers(pbso_group *g, Rank &rk, int nb, int mode) {
    const size_t row = (size_t)nb * g->frames;
    const bool full = mode == PBSO_GATHER_ALL || (mode == PBSO_GATHER_ROOT && rk.rank == 0);
    const size_t want = row * (size_t)(full ? g->world * g->cmax : std::max(1, g->cmax));
    GHIP(g, hipSetDevice(rk.device));
    if (want > rk.target_floats) {
        GHIP(g, hipStreamSynchronize(rk.stream));
        GHIP(g, hipStreamSynchronize(rk.coll));
        for (int s = 0; s < 2; ++s) {
            if (rk.target[s]) GHIP(g, hipFree(rk.target[s]));
            GHIP(g, hipMalloc(&rk.target[s], want * sizeof(float)));
            GHIP(g, hipMemsetAsync(rk.target[s], 0, want * sizeof(float), rk.stream));     // (padding rows of ragged shards stay silent)
        }
        rk.target_floats = want;
    }
    if (row > rk.mix_floats) {
        GHIP(g, hipStreamSynchronize(rk.coll));
        for (int s = 0; s < 2; ++s) {
            if (rk.mix[s]) GHIP(g, hipFree(rk.mix[s]));
            GHIP(g, hipMalloc(&rk.mix[s], row * sizeof(float)));
        }
        rk.mix_floats = row;
    }
    size_t want_scratch = 0;
    if (g->loopback()) want_scratch = row * (size_t)(g->world + pbso::mix_objects_groups(g->world) + 1);
    else if (g->use_rccl && g->world == 1) want_scratch = row * (size_t)std::max(1, g->cmax);
    if (want_scratch > rk.scratch_floats) {
        GHIP(g, hipStreamSynchronize(rk.coll));
        if (rk.scratch) GHIP(g, hipFree(rk.scratch));
        GHIP(g, hipMalloc(&rk.scratch, want_scratch * sizeof(float)));
        rk.scratch_floats = want_scratch;
    }
    return PBSO_OK;
}

}  // namespace

extern "C" {

int pbso_group_unique_id(void *out) {
    if (!out) return PBSO_ERR_INVALID;
    if (!g_rccl.load()) return PBSO_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == PBSO_GROUP_ID_BYTES, "PBSO_GROUP_ID_BYTES");
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return PBSO_ERR_HIP;
    std::memcpy(out, &id, sizeof(id));
    return PBSO_OK;
}

const char *pbso_group_last_error(const pbso_group *g) { return g ? g->err.c_str() : "null group"; }

int pbso_group_create(const pbso_group_desc *d, pbso_group **out) {
    if (!d || !out) return PBSO_ERR_INVALID;
    *out = nullptr;
    pbso_group *g = nullptr;
    try {
        g = new pbso_group();
        *out = g;                                        // (kept alive on failure so that the caller can read the error text)
        if (d->abi_version != PBSO_ABI_VERSION) return gfail(g, PBSO_ERR_INVALID, "abi_version mismatch");
        if (d->n_devices < 1 || !d->devices) return gfail(g, PBSO_ERR_INVALID, "a group needs at least one device");
        g->world = d->world_size > 0 ? d->world_size : d->n_devices;
        g->first = d->first_rank;
        if (g->first < 0 || g->first + d->n_devices > g->world) return gfail(g, PBSO_ERR_INVALID, "first_rank + n_devices exceeds world_size");
        g->transport = d->transport;
        if (g->transport != PBSO_GROUP_RCCL && g->transport != PBSO_GROUP_RCCL_ALWAYS && g->transport != PBSO_GROUP_LOOPBACK)
            return gfail(g, PBSO_ERR_INVALID, "transport");
        if (g->loopback()) {
            // every rank of the job in this process, any of them on any device (the same one, typically): the collectives are
            // device copies -- what the group does AROUND them (shards, padding, slices, the two targets, the events) is the product's
            if (g->world != d->n_devices || g->first != 0) return gfail(g, PBSO_ERR_INVALID, "the loopback transport holds every rank of the job in one process");
        } else {
            if (g->world > d->n_devices && !d->unique_id) return gfail(g, PBSO_ERR_INVALID, "a job of several processes needs the shared unique_id");
            for (int i = 0; i < d->n_devices; ++i)
                for (int j = 0; j < i; ++j)
                    if (d->devices[i] == d->devices[j]) return gfail(g, PBSO_ERR_INVALID, "a device appears twice (one rank per GPU)");
        }
        g->edesc = d->engine;
        g->edesc.abi_version = PBSO_ABI_VERSION;
        g->edesc.submit_thread = 0;          // (the group puts its collectives on the ranks' streams right behind a step: no deferred launches)
        g->frames = d->engine.frames_per_buffer > 0 ? d->engine.frames_per_buffer : PBSO_FRAMES_PER_BUFFER;
        int ndev = 0;
        GHIP(g, hipGetDeviceCount(&ndev));
        if (ndev <= 0) return gfail(g, PBSO_ERR_HIP, "no HIP device: this engine has no CPU fallback");
        g->ranks.resize(d->n_devices);
        for (int i = 0; i < d->n_devices; ++i) {
            Rank &rk = g->ranks[i];
            rk.rank = g->first + i;
            rk.device = d->devices[i];
            if (rk.device < 0 || rk.device >= ndev) return gfail(g, PBSO_ERR_INVALID, "device ordinal out of range");
            GHIP(g, hipSetDevice(rk.device));
            GHIP(g, hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking));
            GHIP(g, hipStreamCreateWithFlags(&rk.coll, hipStreamNonBlocking));
            for (int s = 0; s < 2; ++s) {
                GHIP(g, hipEventCreateWithFlags(&rk.ev_step[s], hipEventDisableTiming));
                GHIP(g, hipEventCreateWithFlags(&rk.ev_coll[s], hipEventDisableTiming));
            }
            pbso_engine_desc ed = g->edesc;
            ed.device = rk.device;
            ed.stream = rk.stream;
            int rc = pbso_engine_create(&ed, &rk.eng);
            if (rc != PBSO_OK) return gfail(g, rc, std::string("engine on device ") + std::to_string(rk.device) + ": " + (rk.eng ? pbso_last_error(rk.eng) : "create failed"));
        }
        g->use_rccl = !g->loopback() && (g->world > 1 || g->transport == PBSO_GROUP_RCCL_ALWAYS);
        if (g->use_rccl) {
            if (!g_rccl.load()) return gfail(g, PBSO_ERR_HIP, g_rccl.err);
            if (g->world == 1) {
                // (tests: a communicator of one rank, created the way a rank of a larger job creates its own)
                ncclUniqueId id;
                if (d->unique_id) std::memcpy(&id, d->unique_id, sizeof(id));
                else GNCCL(g, g_rccl.GetUniqueId(&id));
                GHIP(g, hipSetDevice(g->ranks[0].device));
                GNCCL(g, g_rccl.CommInitRank(&g->ranks[0].comm, 1, id, 0));
            } else if (g->world == d->n_devices) {
                std::vector<ncclComm_t> comms(d->n_devices);
                GNCCL(g, g_rccl.CommInitAll(comms.data(), d->n_devices, d->devices));
                for (int i = 0; i < d->n_devices; ++i) g->ranks[i].comm = comms[i];
            } else {
                ncclUniqueId id;
                std::memcpy(&id, d->unique_id, sizeof(id));
                GNCCL(g, g_rccl.GroupStart());
                for (Rank &rk : g->ranks) {
                    GHIP_OPEN(g, hipSetDevice(rk.device));
                    GNCCL_OPEN(g, g_rccl.CommInitRank(&rk.comm, g->world, id, rk.rank));
                }
                GNCCL(g, g_rccl.GroupEnd());
            }
        }
        return PBSO_OK;
    } catch (const std::exception &ex) {
        return gfail(g, PBSO_ERR_NOMEM, ex.what());
    }
}

void pbso_group_destroy(pbso_group *g) {
    if (!g) return;
    for (Rank &rk : g->ranks) {
        (void)hipSetDevice(rk.device);
        if (rk.stream) (void)hipStreamSynchronize(rk.stream);
        if (rk.coll) (void)hipStreamSynchronize(rk.coll);
        if (rk.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(rk.comm);
        if (rk.eng) pbso_engine_destroy(rk.eng);
        for (int s = 0; s < 2; ++s) {
            if (rk.target[s]) (void)hipFree(rk.target[s]);
            if (rk.mix[s]) (void)hipFree(rk.mix[s]);
            if (s == 0 && rk.scratch) (void)hipFree(rk.scratch);
            if (rk.ev_step[s]) (void)hipEventDestroy(rk.ev_step[s]);
            if (rk.ev_coll[s]) (void)hipEventDestroy(rk.ev_coll[s]);
        }
        if (rk.coll) (void)hipStreamDestroy(rk.coll);
        if (rk.stream) (void)hipStreamDestroy(rk.stream);
    }
    delete g;
}

// shards balanced by the sum of modes: the cut points are the object boundaries nearest to the ideal prefix sums
// k / world of the total (the same rule as openpbso_amd/distributed.py shard_by_modes: every rank computes the same cuts)
int pbso_shard_by_modes(const int *modes, int n, int world, int *cuts) {
    if (n < 0 || world < 1 || !cuts || (n > 0 && !modes)) return PBSO_ERR_INVALID;
    std::vector<long long> prefix(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (modes[i] < 0) return PBSO_ERR_INVALID;
        prefix[i + 1] = prefix[i] + modes[i];
    }
    const long long total = prefix[n];
    cuts[0] = 0;
    for (int k = 1; k < world; ++k) {
        int c;
        if (total == 0) {
            const int base = n / world, rem = n % world;
            c = k * base + std::min(k, rem);
        } else {
            const double target = (double)total * k / world;
            c = (int)(std::lower_bound(prefix.begin() + cuts[k - 1], prefix.end(), target,
                                       [](long long p, double t) { return (double)p < t; }) - prefix.begin());
            c = std::min(c, n);
            if (c > cuts[k - 1] && target - (double)prefix[c - 1] < (double)prefix[c] - target) c -= 1;
        }
        cuts[k] = std::max(c, cuts[k - 1]);
    }
    cuts[world] = n;
    return PBSO_OK;
}

int pbso_group_plan(pbso_group *g, const int *modes, int n) {
    if (!g || n < 0 || (n > 0 && !modes)) return gfail(g, PBSO_ERR_INVALID, "plan arguments");
    if (g->planned) return gfail(g, PBSO_ERR_STATE, "plan called twice");
    g->cuts.assign((size_t)g->world + 1, 0);
    if (pbso_shard_by_modes(modes, n, g->world, g->cuts.data()) != PBSO_OK) return gfail(g, PBSO_ERR_INVALID, "negative mode count");
    g->n_objects = n;
    g->cmax = 0;
    for (int r = 0; r < g->world; ++r) g->cmax = std::max(g->cmax, g->cuts[r + 1] - g->cuts[r]);
    for (Rank &rk : g->ranks) rk.n_local = g->cuts[rk.rank + 1] - g->cuts[rk.rank];
    g->planned = true;
    return PBSO_OK;
}

int pbso_group_rank_span(pbso_group *g, int rank, int *lo, int *hi) {
    if (!g || !g->planned || rank < 0 || rank >= g->world) return gfail(g, PBSO_ERR_INVALID, "rank_span arguments (plan first)");
    if (lo) *lo = g->cuts[rank];
    if (hi) *hi = g->cuts[rank + 1];
    return PBSO_OK;
}

int pbso_group_owner(pbso_group *g, int id, int *rank, int *local_id) {
    if (!g || !g->planned || id < 0 || id >= g->n_objects) return gfail(g, PBSO_ERR_INVALID, "owner arguments (plan first)");
    const int r = (int)(std::upper_bound(g->cuts.begin(), g->cuts.end(), id) - g->cuts.begin()) - 1;
    if (rank) *rank = r;
    if (local_id) *local_id = id - g->cuts[r];
    return PBSO_OK;
}

int pbso_group_add_object(pbso_group *g, int id, const pbso_object_desc *d) {
    if (!g || !d) return PBSO_ERR_INVALID;
    int r = 0, l = 0;
    int rc = pbso_group_owner(g, id, &r, &l);
    if (rc != PBSO_OK) return rc;
    Rank *rk = local_rank(g, r);
    if (!rk) return PBSO_OK;                              // another process builds it
    if (l != rk->next_local) return gfail(g, PBSO_ERR_STATE, "objects of a rank are added in ascending id order");
    int got = -1;
    GENG(g, *rk, pbso_add_object(rk->eng, d, &got));
    if (got != l) return gfail(g, PBSO_ERR_STATE, "engine numbered the object differently");
    rk->next_local += 1;
    return PBSO_OK;
}

int pbso_group_finalize(pbso_group *g) {
    if (!g || !g->planned) return gfail(g, PBSO_ERR_STATE, "finalize before plan");
    for (Rank &rk : g->ranks) {
        if (rk.next_local != rk.n_local) return gfail(g, PBSO_ERR_STATE, "rank " + std::to_string(rk.rank) + " is missing objects");
        if (rk.n_local > 0) GENG(g, rk, pbso_finalize(rk.eng));
    }
    g->finalized = true;
    return PBSO_OK;
}

pbso_engine *pbso_group_engine(pbso_group *g, int rank) {
    Rank *rk = g ? local_rank(g, rank) : nullptr;
    return rk ? rk->eng : nullptr;
}

int pbso_group_enqueue_force(pbso_group *g, int id, const pbso_force_msg *m, int64_t not_before) {
    int r = 0, l = 0;
    int rc = pbso_group_owner(g, id, &r, &l);
    if (rc != PBSO_OK) return rc;
    Rank *rk = local_rank(g, r);
    if (!rk) return 1;
    int took = pbso_enqueue_force(rk->eng, l, m, not_before);
    if (took < 0) return gfail(g, took, pbso_last_error(rk->eng));
    return took;
}

int pbso_group_step(pbso_group *g, int nb) {
    if (!g || !g->finalized) return gfail(g, PBSO_ERR_STATE, "step before finalize");
    if (nb <= 0) return gfail(g, PBSO_ERR_INVALID, "n_buffers must be > 0");
    const int slot = g->slot;
    const size_t row = (size_t)nb * g->frames;
    for (Rank &rk : g->ranks) {
        // (the target doubles as the all-gather's receive buffer: sized for that from the start, so the engine's slice never moves)
        int rc = ensure_buffers(g, rk, nb, PBSO_GATHER_ALL);
        if (rc != PBSO_OK) return rc;
        GHIP(g, hipSetDevice(rk.device));
        // the collective that last read this target is done (loopback: the OTHER ranks' copies read this rank's slice)
        if (g->loopback()) for (Rank &other : g->ranks) GHIP(g, hipStreamWaitEvent(rk.stream, other.ev_coll[slot], 0));
        else GHIP(g, hipStreamWaitEvent(rk.stream, rk.ev_coll[slot], 0));
        float *mine = rk.target[slot] + (size_t)rk.rank * g->cmax * row;
        // Ragged shards: the rows behind this rank's objects are shipped to every rank with the all-gather and must be SILENT.
        // The row layout depends on n_buffers, so a step shorter than an earlier one would find old samples there: zeroed
        // every step (a few rows), on the stream the step runs on.
        if (rk.n_local < g->cmax)
            GHIP(g, hipMemsetAsync(mine + (size_t)rk.n_local * row, 0, (size_t)(g->cmax - rk.n_local) * row * sizeof(float), rk.stream));
        if (rk.n_local > 0) GENG(g, rk, pbso_step_into(rk.eng, nb, mine));
        GHIP(g, hipEventRecord(rk.ev_step[slot], rk.stream));       // (an empty rank too: its silent rows are ordered before the collective)
    }
    g->last_slot = slot;
    g->last_nb = nb;
    g->last_mode = 0;
    g->slot ^= 1;
    g->stepped = true;
    return PBSO_OK;
}

// point-to-point transfers in pieces of at most 256 MB: a 1.8 GB ncclSend / ncclRecv pair of a rank to itself (1024 objects x ten seconds
// of audio) came back WRONG on RCCL 2.26.6 while 0.45 GB was right (profiles/r05 bench_1rank.err); sender and receiver cut alike
static constexpr size_t P2P_MAX = (size_t)64 << 20;      // floats

int pbso_group_gather(pbso_group *g, int mode) {
    if (!g || !g->stepped) return gfail(g, PBSO_ERR_STATE, "gather before step");
    if (mode != PBSO_GATHER_ALL && mode != PBSO_GATHER_ROOT && mode != PBSO_GATHER_MIX) return gfail(g, PBSO_ERR_INVALID, "gather mode");
    const int slot = g->last_slot;
    const size_t row = (size_t)g->last_nb * g->frames, blk = (size_t)g->cmax * row;
    if (mode == PBSO_GATHER_MIX) {
        for (Rank &rk : g->ranks) {
            GHIP(g, hipSetDevice(rk.device));
            // (a second collective on the same step: the first one's reads of the mix row are done)
            if (g->loopback()) for (Rank &other : g->ranks) GHIP(g, hipStreamWaitEvent(rk.stream, other.ev_coll[slot], 0));
            else GHIP(g, hipStreamWaitEvent(rk.stream, rk.ev_coll[slot], 0));
            if (rk.n_local > 0) GENG(g, rk, pbso_mix_objects(rk.eng, rk.mix[slot]));
            else GHIP(g, hipMemsetAsync(rk.mix[slot], 0, row * sizeof(float), rk.stream));
            GHIP(g, hipEventRecord(rk.ev_step[slot], rk.stream));
        }
    }
    for (Rank &rk : g->ranks) {
        GHIP(g, hipSetDevice(rk.device));
        // (loopback: a rank's copies read the other ranks' slices)
        if (g->loopback()) for (Rank &other : g->ranks) GHIP(g, hipStreamWaitEvent(rk.coll, other.ev_step[slot], 0));
        else GHIP(g, hipStreamWaitEvent(rk.coll, rk.ev_step[slot], 0));
    }
    if (g->use_rccl) {
        GNCCL(g, g_rccl.GroupStart());
        for (Rank &rk : g->ranks) {
            float *base = rk.target[slot], *mine = base + (size_t)rk.rank * blk;
            if (mode == PBSO_GATHER_ALL) {
                GNCCL_OPEN(g, g_rccl.AllGather(mine, base, blk, ncclFloat, rk.comm, rk.coll));          // in place: sendbuff == recvbuff + rank * count
            } else if (mode == PBSO_GATHER_MIX) {
                GNCCL_OPEN(g, g_rccl.AllReduce(rk.mix[slot], rk.mix[slot], row, ncclFloat, ncclSum, rk.comm, rk.coll));
            } else if (g->world == 1) {
                // (PBSO_GROUP_RCCL_ALWAYS: the root's receive and a rank's send, both on the one rank there is -- the rows land in
                //  the scratch rows, which is what pbso_group_result_device_ptr then hands out)
                for (size_t o = 0; o < blk; o += P2P_MAX) {
                    const size_t n = std::min(P2P_MAX, blk - o);
                    GNCCL_OPEN(g, g_rccl.Send(mine + o, n, ncclFloat, 0, rk.comm, rk.coll));
                    GNCCL_OPEN(g, g_rccl.Recv(rk.scratch + o, n, ncclFloat, 0, rk.comm, rk.coll));
                }
            } else if (rk.rank == 0) {
                for (int r = 1; r < g->world; ++r)
                    for (size_t o = 0; o < blk; o += P2P_MAX)
                        GNCCL_OPEN(g, g_rccl.Recv(base + (size_t)r * blk + o, std::min(P2P_MAX, blk - o), ncclFloat, r, rk.comm, rk.coll));
            } else {
                for (size_t o = 0; o < blk; o += P2P_MAX)
                    GNCCL_OPEN(g, g_rccl.Send(mine + o, std::min(P2P_MAX, blk - o), ncclFloat, 0, rk.comm, rk.coll));
            }
        }
        GNCCL(g, g_rccl.GroupEnd());
    } else if (g->loopback()) {
        for (Rank &rk : g->ranks) {
            GHIP(g, hipSetDevice(rk.device));
            if (mode == PBSO_GATHER_ALL || (mode == PBSO_GATHER_ROOT && rk.rank == 0)) {
                for (Rank &src : g->ranks) {
                    if (src.rank == rk.rank) continue;
                    GHIP(g, hipMemcpyAsync(rk.target[slot] + (size_t)src.rank * blk, src.target[slot] + (size_t)src.rank * blk, blk * sizeof(float),
                                           hipMemcpyDeviceToDevice, rk.coll));
                }
            } else if (mode == PBSO_GATHER_MIX) {
                // all-reduce: every rank collects the ranks' mixed rows and adds them in rank order (the object sum's kernels)
                float *rows = rk.scratch, *parts = rk.scratch + (size_t)g->world * row;
                for (Rank &src : g->ranks)
                    GHIP(g, hipMemcpyAsync(rows + (size_t)src.rank * row, src.mix[slot], row * sizeof(float), hipMemcpyDeviceToDevice, rk.coll));
                // (into a row of its own: the other ranks' copies still read this rank's mix row)
                int lrc = pbso::launch_mix_objects(rows, g->world, (long long)row, (long long)row, parts, loop_mix_out(g, rk, row), rk.coll);
                if (lrc != 0) return gfail(g, PBSO_ERR_HIP, "loopback all-reduce: launch_mix_objects");
            }
        }
    }
    for (Rank &rk : g->ranks) {
        GHIP(g, hipSetDevice(rk.device));
        GHIP(g, hipEventRecord(rk.ev_coll[slot], rk.coll));
    }
    g->last_mode = mode;
    return PBSO_OK;
}

int pbso_group_sync(pbso_group *g) {
    if (!g) return PBSO_ERR_INVALID;
    for (Rank &rk : g->ranks) {
        GHIP(g, hipSetDevice(rk.device));
        if (rk.eng && rk.n_local > 0) GENG(g, rk, pbso_sync(rk.eng));
        GHIP(g, hipStreamSynchronize(rk.stream));
        GHIP(g, hipStreamSynchronize(rk.coll));
    }
    return PBSO_OK;
}

void *pbso_group_result_device_ptr(pbso_group *g, int rank, size_t *rows, size_t *row_floats) {
    Rank *rk = g ? local_rank(g, rank) : nullptr;
    if (!rk || g->last_slot < 0) return nullptr;
    const size_t row = (size_t)g->last_nb * g->frames;
    if (row_floats) *row_floats = row;
    if (g->last_mode == PBSO_GATHER_MIX) {
        if (rows) *rows = 1;
        return g->loopback() ? loop_mix_out(g, *rk, row) : rk->mix[g->last_slot];
    }
    const bool full = g->last_mode == PBSO_GATHER_ALL || (g->last_mode == PBSO_GATHER_ROOT && rk->rank == 0);
    if (rows) *rows = full ? (size_t)g->world * g->cmax : (size_t)g->cmax;
    // (a one-rank communicator made for tests: the rows this rank sent to itself through ncclSend / ncclRecv)
    if (g->last_mode == PBSO_GATHER_ROOT && g->use_rccl && g->world == 1) return rk->scratch;
    return full ? rk->target[g->last_slot] : rk->target[g->last_slot] + (size_t)rk->rank * g->cmax * row;
}

int pbso_group_read_result(pbso_group *g, int rank, float *out, size_t n) {
    size_t rows = 0, row = 0;
    void *p = pbso_group_result_device_ptr(g, rank, &rows, &row);
    if (!p || !out) return gfail(g, PBSO_ERR_STATE, "no result on that rank");
    if (n != rows * row) return gfail(g, PBSO_ERR_INVALID, "read_result size mismatch");
    int rc = pbso_group_sync(g);
    if (rc != PBSO_OK) return rc;
    Rank *rk = local_rank(g, rank);
    GHIP(g, hipSetDevice(rk->device));
    GHIP(g, hipMemcpy(out, p, n * sizeof(float), hipMemcpyDeviceToHost));
    return PBSO_OK;
}

}  // extern "C"
