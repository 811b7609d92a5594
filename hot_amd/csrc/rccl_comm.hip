// libhotmi355x — a native implementation of hot_comm (include/hot_mi355x.h) on RCCL: the three collectives of the sharded
// solve as ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv enqueued on the CONTEXT'S OWN HIP STREAM.  Stream-ordered:
// no host synchronisation and no interpreter in the loop — a colour of a Gauss-Seidel sweep is "kernel, pack, all-gather, unpack"
// back to back on one stream.  (hot_amd/dist.py's TorchComm does the same through torch.distributed with host
// synchronisation on both sides; it is the reference implementation the multi-rank tests run, over gloo on CPU / one GPU.)
//
// RCCL is resolved with dlopen at run time, so the library loads on hosts without it and hot_rccl_* then report an error
// (callers fall back to TorchComm).  Bootstrap: rank 0 obtains hot_rccl_unique_id(), the host distributes the 128 bytes by
// whatever means it has (bench.py: torch.distributed broadcast), every rank calls hot_rccl_attach.
#include "hot_ctx.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};
Api& api()
{
    static Api a;
    if (a.lib || a.ok) return a;
    for (const char* name : { "librccl.so.1", "librccl.so" }) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) return a;
#define HOT_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, sym))
    HOT_SYM(GetUniqueId, "ncclGetUniqueId"), HOT_SYM(CommInitRank, "ncclCommInitRank"), HOT_SYM(CommDestroy, "ncclCommDestroy"), HOT_SYM(AllReduce, "ncclAllReduce");
    HOT_SYM(AllGather, "ncclAllGather"), HOT_SYM(Send, "ncclSend"), HOT_SYM(Recv, "ncclRecv"), HOT_SYM(GroupStart, "ncclGroupStart"), HOT_SYM(GroupEnd, "ncclGroupEnd");
#undef HOT_SYM
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
    return a;
}

struct Rccl {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, size = 1;
    bool self_via_p2p = false; // self-test: a rank's message to itself goes through the grouped ncclSend / ncclRecv as well
    char* scratch = nullptr; // device staging of host payloads (scalars, counts)
    size_t scratch_bytes = 0;
    char* stage(size_t bytes)
    {
        if (bytes > scratch_bytes) {
            if (scratch) (void)hipFree(scratch);
            scratch_bytes = bytes < 4096 ? 4096 : 2 * bytes;
            if (hipMalloc((void**)&scratch, scratch_bytes) != hipSuccess) scratch = nullptr, scratch_bytes = 0;
        }
        return scratch;
    }
};

ncclDataType_t dtype_of(int32_t d) { return d == HOT_COMM_F32 ? ncclFloat : d == HOT_COMM_F64 ? ncclDouble : d == HOT_COMM_I32 ? ncclInt32 : ncclInt64; }
size_t size_of(int32_t d) { return (d == HOT_COMM_F32 || d == HOT_COMM_I32) ? 4 : 8; }

int32_t cb_allreduce(void* user, void* buf, int64_t n, int32_t dtype, int32_t op, int32_t on_device)
{
    Rccl* r = (Rccl*)user;
    const ncclRedOp_t rop = op == HOT_COMM_MAX ? ncclMax : ncclSum;
    if (on_device) return api().AllReduce(buf, buf, (size_t)n, dtype_of(dtype), rop, r->comm, r->stream) == ncclSuccess ? 0 : 1;
    // host payload: the caller needs the result on the host now
    const size_t bytes = (size_t)n * size_of(dtype);
    char* d = r->stage(bytes);
    if (!d) return 1;
    if (hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 1;
    if (api().AllReduce(d, d, (size_t)n, dtype_of(dtype), rop, r->comm, r->stream) != ncclSuccess) return 1;
    if (hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 1;
    return hipStreamSynchronize(r->stream) == hipSuccess ? 0 : 1;
}
int32_t cb_allgather(void* user, const void* send, void* recv, int64_t bytes, int32_t on_device)
{
    Rccl* r = (Rccl*)user;
    if (on_device) return api().AllGather(send, recv, (size_t)bytes, ncclChar, r->comm, r->stream) == ncclSuccess ? 0 : 1;
    char* d = r->stage((size_t)bytes * (r->size + 1));
    if (!d) return 1;
    if (hipMemcpyAsync(d, send, (size_t)bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 1;
    if (api().AllGather(d, d + bytes, (size_t)bytes, ncclChar, r->comm, r->stream) != ncclSuccess) return 1;
    if (hipMemcpyAsync(recv, d + bytes, (size_t)bytes * r->size, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 1;
    return hipStreamSynchronize(r->stream) == hipSuccess ? 0 : 1;
}
int32_t cb_alltoallv(void* user, const void* send, const int64_t* soff, const int64_t* sbytes, void* recv, const int64_t* roff, const int64_t* rbytes, int32_t on_device)
{
    Rccl* r = (Rccl*)user;
    if (!on_device) return 1; // the library only exchanges device payloads this way
    bool ok = true;
    if (sbytes[r->rank] > 0 && !r->self_via_p2p) // a rank's message to itself (the library sends none today; the contract allows it)
        ok = hipMemcpyAsync((char*)recv + roff[r->rank], (const char*)send + soff[r->rank], (size_t)sbytes[r->rank], hipMemcpyDeviceToDevice, r->stream) == hipSuccess;
    if (!ok || api().GroupStart() != ncclSuccess) return 1; // no GroupEnd without a GroupStart
    for (int p = 0; p < r->size && ok; ++p) {
        if (p == r->rank && !r->self_via_p2p) continue; // (self_via_p2p, self-test only: the message to itself takes the grouped Send / Recv like a peer's)
        if (rbytes[p] > 0) ok = ok && api().Recv((char*)recv + roff[p], (size_t)rbytes[p], ncclChar, p, r->comm, r->stream) == ncclSuccess;
        if (sbytes[p] > 0) ok = ok && api().Send((const char*)send + soff[p], (size_t)sbytes[p], ncclChar, p, r->comm, r->stream) == ncclSuccess;
    }
    return (api().GroupEnd() == ncclSuccess && ok) ? 0 : 1;
}

} // namespace

struct hot_ctx {
    hot::CtxBase* impl = nullptr;
    std::string err;
};

extern "C" {

int hot_rccl_unique_id(void* out128)
{
    if (!out128 || !api().ok) return HOT_ERR_DEVICE;
    ncclUniqueId id;
    if (api().GetUniqueId(&id) != ncclSuccess) return HOT_ERR_DEVICE;
    std::memcpy(out128, &id, sizeof(id));
    return HOT_OK;
}

int hot_rccl_attach(hot_ctx* ctx, const void* unique_id128, int32_t rank, int32_t size, int32_t partition_min_rows)
{
    if (!ctx || !ctx->impl || !unique_id128 || size < 1 || rank < 0 || rank >= size) return HOT_ERR_INVALID;
    if (!api().ok) {
        ctx->err = "RCCL is not available (librccl.so could not be loaded)";
        return HOT_ERR_DEVICE;
    }
    (void)hipSetDevice(ctx->impl->cfg.device);
    if (ctx->impl->native_comm) { // a second attach: the previous communicator (and its scratch) goes first — nothing may still be enqueued on it
        (void)hipStreamSynchronize(ctx->impl->stream);
        try {
            ctx->impl->set_comm(nullptr);
        }
        catch (...) {
        }
        if (ctx->impl->native_comm_free) ctx->impl->native_comm_free(ctx->impl->native_comm);
        ctx->impl->native_comm = nullptr, ctx->impl->native_comm_free = nullptr;
    }
    Rccl* r = new (std::nothrow) Rccl;
    if (!r) return HOT_ERR_DEVICE;
    r->rank = rank, r->size = size, r->stream = ctx->impl->stream;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    if (api().CommInitRank(&r->comm, size, id, rank) != ncclSuccess) {
        delete r;
        ctx->err = "ncclCommInitRank failed";
        return HOT_ERR_DEVICE;
    }
    hot_comm c;
    std::memset(&c, 0, sizeof(c));
    c.rank = rank, c.size = size, c.user = r;
    c.allreduce = cb_allreduce, c.allgather = cb_allgather, c.alltoallv = cb_alltoallv;
    c.partition_min_rows = partition_min_rows;
    c.stream_ordered = 1;
    try {
        ctx->impl->set_comm(&c);
    }
    catch (const hot::Error& e) {
        ctx->err = e.msg;
        (void)api().CommDestroy(r->comm);
        delete r;
        return e.code;
    }
    catch (const std::exception& e) { // nothing may cross the extern "C" boundary
        ctx->err = e.what();
        (void)api().CommDestroy(r->comm);
        delete r;
        return HOT_ERR_DEVICE;
    }
    catch (...) {
        ctx->err = "hot_rccl_attach: unknown exception";
        (void)api().CommDestroy(r->comm);
        delete r;
        return HOT_ERR_DEVICE;
    }
    ctx->impl->native_comm = r;
    ctx->impl->native_comm_free = [](void* p) {
        Rccl* q = (Rccl*)p;
        if (q->comm) (void)api().CommDestroy(q->comm);
        if (q->scratch) (void)hipFree(q->scratch);
        delete q;
    };
    return HOT_OK;
}

// Drives every callback of the attached communicator with known data (device and host payloads) and checks the results: all-gather,
// all-reduce (sum, max), and a personalised exchange in which every (source, destination) pair has its own length and contents.  On the
// one-GPU test box a group has a single rank: the exchange then consists of the rank's message to itself, sent once by copy and once through
// the grouped ncclSend / ncclRecv.  bench.py --gpus N runs it on every rank before the first step.
int hot_rccl_selftest(hot_ctx* ctx)
{
    if (!ctx || !ctx->impl || !ctx->impl->native_comm) return HOT_ERR_INVALID;
    Rccl* r = (Rccl*)ctx->impl->native_comm;
    (void)hipSetDevice(ctx->impl->cfg.device);
    const int n = 1000;
    std::vector<double> h(n), back(n * (size_t)r->size);
    for (int i = 0; i < n; ++i) h[i] = i + 0.5 * r->rank;
    double* d = nullptr;
    double* g = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(double)) != hipSuccess || hipMalloc((void**)&g, n * sizeof(double) * r->size) != hipSuccess) return HOT_ERR_DEVICE;
    bool ok = hipMemcpyAsync(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice, r->stream) == hipSuccess;
    ok = ok && cb_allgather(r, d, g, n * (int64_t)sizeof(double), 1) == 0; // stream-ordered after the copy
    ok = ok && cb_allreduce(r, d, n, HOT_COMM_F64, HOT_COMM_SUM, 1) == 0;
    ok = ok && hipMemcpyAsync(back.data(), g, n * sizeof(double) * r->size, hipMemcpyDeviceToHost, r->stream) == hipSuccess;
    std::vector<double> sum(n);
    ok = ok && hipMemcpyAsync(sum.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, r->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(r->stream) == hipSuccess;
    for (int i = 0; i < n && ok; ++i) {
        ok = back[(size_t)r->rank * n + i] == h[i]; // my slot of the all-gather is my data
        double expect = 0;
        for (int q = 0; q < r->size; ++q) expect += i + 0.5 * q;
        ok = ok && sum[i] == expect;
    }
    double hs[3] = { 1.0 + r->rank, 2.0, -3.0 };
    ok = ok && cb_allreduce(r, hs, 3, HOT_COMM_F64, HOT_COMM_MAX, 0) == 0 && hs[0] == (double)r->size && hs[1] == 2.0 && hs[2] == -3.0;
    int64_t cnt = 40 + r->rank;
    std::vector<int64_t> all(r->size, -1);
    ok = ok && cb_allgather(r, &cnt, all.data(), sizeof(int64_t), 0) == 0;
    for (int q = 0; q < r->size && ok; ++q) ok = all[q] == 40 + q;
    // personalised exchange, every (source, destination) pair — a rank's message to itself included — with its own length and contents:
    // rank s sends len(s, p) = 8 (1 + (s + 2 p) % 5) doubles  1e6 s + 1e3 p + k  to rank p.  Run twice: the message to itself by the
    // stream-ordered copy (what the library's calls take), then through the grouped ncclSend / ncclRecv like a peer's, so that the
    // Send / Recv loop has carried real, checked traffic on a one-rank group too.
    auto len = [](int s_, int p_) { return (int64_t)8 * (1 + (s_ + 2 * p_) % 5); };
    std::vector<int64_t> so(r->size, 0), sb(r->size, 0), ro(r->size, 0), rb(r->size, 0);
    int64_t stot = 0, rtot = 0;
    for (int p = 0; p < r->size; ++p) so[p] = stot * 8, sb[p] = len(r->rank, p) * 8, stot += len(r->rank, p), ro[p] = rtot * 8, rb[p] = len(p, r->rank) * 8, rtot += len(p, r->rank);
    std::vector<double> hsend(stot), hrecv(rtot);
    for (int p = 0; p < r->size; ++p)
        for (int64_t k = 0; k < len(r->rank, p); ++k) hsend[so[p] / 8 + k] = 1e6 * r->rank + 1e3 * p + (double)k;
    double *xs = nullptr, *xr = nullptr;
    ok = ok && hipMalloc((void**)&xs, stot * sizeof(double)) == hipSuccess && hipMalloc((void**)&xr, rtot * sizeof(double)) == hipSuccess;
    for (int pass = 0; pass < 2 && ok; ++pass) {
        r->self_via_p2p = pass == 1;
        ok = hipMemcpyAsync(xs, hsend.data(), stot * sizeof(double), hipMemcpyHostToDevice, r->stream) == hipSuccess && hipMemsetAsync(xr, 0xff, rtot * sizeof(double), r->stream) == hipSuccess;
        ok = ok && cb_alltoallv(r, xs, so.data(), sb.data(), xr, ro.data(), rb.data(), 1) == 0;
        ok = ok && hipMemcpyAsync(hrecv.data(), xr, rtot * sizeof(double), hipMemcpyDeviceToHost, r->stream) == hipSuccess && hipStreamSynchronize(r->stream) == hipSuccess;
        for (int p = 0; p < r->size && ok; ++p)
            for (int64_t k = 0; k < len(p, r->rank) && ok; ++k) ok = hrecv[ro[p] / 8 + k] == 1e6 * p + 1e3 * r->rank + (double)k;
    }
    r->self_via_p2p = false;
    if (xs) (void)hipFree(xs);
    if (xr) (void)hipFree(xr);
    (void)hipFree(d), (void)hipFree(g);
    if (!ok) ctx->err = "hot_rccl_selftest: a collective returned an error or a wrong result";
    return ok ? HOT_OK : HOT_ERR_DEVICE;
}

} // extern "C"
