// Device group: one engine per GPU, objects sharded over the ranks, RCCL only to gather the finished audio
// (include/openpbso_amd.h "device group").  Objects never interact (modal_solver.h:100-126), so
// stepping needs no exchange; this file is host-side plumbing over the C ABI of the single engine plus three collectives.
// librccl is loaded at run time (dlopen) when a group of more than one rank is created: an engine alone does not need it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/openpbso_amd.h"
#include "kernels.h"

namespace {

struct Rccl {
    void *lib = nullptr;
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
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("librccl not found: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](auto &fp, const char *n) {
            fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(dlsym(lib, n));
            if (!fp) { ok = false; err = std::string("librccl lacks ") + n; }
        };
        sym(GetUniqueId, "ncclGetUniqueId"); sym(CommInitRank, "ncclCommInitRank"); sym(CommInitAll, "ncclCommInitAll");
        sym(CommDestroy, "ncclCommDestroy"); sym(AllGather, "ncclAllGather"); sym(AllReduce, "ncclAllReduce");
        sym(Send, "ncclSend"); sym(Recv, "ncclRecv"); sym(GroupStart, "ncclGroupStart"); sym(GroupEnd, "ncclGroupEnd");
        sym(GetErrorString, "ncclGetErrorString");
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
    size_t target_floats = 0, mix_floats = 0;
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
#define GENG(g, rk, expr)                                                                         \
    do {                                                                                          \
        int _c = (expr);                                                                          \
        if (_c < 0) return gfail(g, _c, std::string("rank ") + std::to_string((rk).rank) + ": " + pbso_last_error((rk).eng)); \
    } while (0)

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
        if (g->world > d->n_devices && !d->unique_id) return gfail(g, PBSO_ERR_INVALID, "a job of several processes needs the shared unique_id");
        for (int i = 0; i < d->n_devices; ++i)
            for (int j = 0; j < i; ++j)
                if (d->devices[i] == d->devices[j]) return gfail(g, PBSO_ERR_INVALID, "a device appears twice (one rank per GPU)");
        g->edesc = d->engine;
        g->edesc.abi_version = PBSO_ABI_VERSION;
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
        if (g->world > 1) {
            if (!g_rccl.load()) return gfail(g, PBSO_ERR_HIP, g_rccl.err);
            if (g->world == d->n_devices) {
                std::vector<ncclComm_t> comms(d->n_devices);
                GNCCL(g, g_rccl.CommInitAll(comms.data(), d->n_devices, d->devices));
                for (int i = 0; i < d->n_devices; ++i) g->ranks[i].comm = comms[i];
            } else {
                ncclUniqueId id;
                std::memcpy(&id, d->unique_id, sizeof(id));
                GNCCL(g, g_rccl.GroupStart());
                for (Rank &rk : g->ranks) {
                    GHIP(g, hipSetDevice(rk.device));
                    GNCCL(g, g_rccl.CommInitRank(&rk.comm, g->world, id, rk.rank));
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
        if (rk.n_local == 0) continue;
        GHIP(g, hipSetDevice(rk.device));
        GHIP(g, hipStreamWaitEvent(rk.stream, rk.ev_coll[slot], 0));      // the collective that last read this target is done
        GENG(g, rk, pbso_step_into(rk.eng, nb, rk.target[slot] + (size_t)rk.rank * g->cmax * row));
        GHIP(g, hipEventRecord(rk.ev_step[slot], rk.stream));
    }
    g->last_slot = slot;
    g->last_nb = nb;
    g->last_mode = 0;
    g->slot ^= 1;
    g->stepped = true;
    return PBSO_OK;
}

int pbso_group_gather(pbso_group *g, int mode) {
    if (!g || !g->stepped) return gfail(g, PBSO_ERR_STATE, "gather before step");
    if (mode != PBSO_GATHER_ALL && mode != PBSO_GATHER_ROOT && mode != PBSO_GATHER_MIX) return gfail(g, PBSO_ERR_INVALID, "gather mode");
    const int slot = g->last_slot;
    const size_t row = (size_t)g->last_nb * g->frames, blk = (size_t)g->cmax * row;
    if (mode == PBSO_GATHER_MIX) {
        for (Rank &rk : g->ranks) {
            GHIP(g, hipSetDevice(rk.device));
            if (rk.n_local > 0) GENG(g, rk, pbso_mix_objects(rk.eng, rk.mix[slot]));
            else GHIP(g, hipMemsetAsync(rk.mix[slot], 0, row * sizeof(float), rk.stream));
            GHIP(g, hipEventRecord(rk.ev_step[slot], rk.stream));
        }
    }
    for (Rank &rk : g->ranks) {
        GHIP(g, hipSetDevice(rk.device));
        GHIP(g, hipStreamWaitEvent(rk.coll, rk.ev_step[slot], 0));
    }
    if (g->world > 1) {
        GNCCL(g, g_rccl.GroupStart());
        for (Rank &rk : g->ranks) {
            float *base = rk.target[slot], *mine = base + (size_t)rk.rank * blk;
            if (mode == PBSO_GATHER_ALL) {
                GNCCL(g, g_rccl.AllGather(mine, base, blk, ncclFloat, rk.comm, rk.coll));          // in place: sendbuff == recvbuff + rank * count
            } else if (mode == PBSO_GATHER_MIX) {
                GNCCL(g, g_rccl.AllReduce(rk.mix[slot], rk.mix[slot], row, ncclFloat, ncclSum, rk.comm, rk.coll));
            } else if (rk.rank == 0) {
                for (int r = 1; r < g->world; ++r) GNCCL(g, g_rccl.Recv(base + (size_t)r * blk, blk, ncclFloat, r, rk.comm, rk.coll));
            } else {
                GNCCL(g, g_rccl.Send(mine, blk, ncclFloat, 0, rk.comm, rk.coll));
            }
        }
        GNCCL(g, g_rccl.GroupEnd());
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
        return rk->mix[g->last_slot];
    }
    const bool full = g->last_mode == PBSO_GATHER_ALL || (g->last_mode == PBSO_GATHER_ROOT && rk->rank == 0);
    if (rows) *rows = full ? (size_t)g->world * g->cmax : (size_t)g->cmax;
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
