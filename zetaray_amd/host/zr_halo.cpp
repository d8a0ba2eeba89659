// zr_halo.cpp -- the reservoir halo exchange of the screen-tile split as a RenderGraph node, in C++ over RCCL.
//
// Not in the reference (single GPU).  One frame of a tiled ReSTIR pass runs  temporal stage -> HaloExchange(POST_TEMPORAL) -> spatial stage
// -> HaloExchange(FINAL): each rank packs the border strips every neighbour needs into ONE send buffer with one kernel
// (zr_pass_halo_pack_all), posts ncclSend / ncclRecv for all peers inside one ncclGroupStart / ncclGroupEnd on the pass's stream, and
// scatters the received strips into its apron with one more kernel -- three stream operations, no host wait, nothing reduced (xGMI is
// point-to-point: every neighbour pair has its own link, so grouped P2P is the natural collective here).
// RCCL is loaded with dlopen so the library still loads on machines without it; a world of one rank exchanges with itself through RCCL as
// well (the self-test of the transport on a single GPU).
#include "zr_host.h"
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
struct Rccl
{
    void* so = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool Load(std::string& err)
    {
        if (so) return true;
        // the instance the process already uses (torch.distributed's) first: two RCCL copies in one process would each run their own bootstrap
        for (const char* name : {"librccl.so.1", "librccl.so"}) { so = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (so) break; }
        if (!so) for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { so = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
        if (!so) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
#define ZR_SYM(field, sym) field = (decltype(field))dlsym(so, sym); if (!field) { err = std::string("librccl.so lacks ") + sym; return false; }
        ZR_SYM(GetUniqueId, "ncclGetUniqueId") ZR_SYM(CommInitRank, "ncclCommInitRank") ZR_SYM(CommDestroy, "ncclCommDestroy")
        ZR_SYM(GroupStart, "ncclGroupStart") ZR_SYM(GroupEnd, "ncclGroupEnd") ZR_SYM(Send, "ncclSend") ZR_SYM(Recv, "ncclRecv")
        ZR_SYM(GetErrorString, "ncclGetErrorString")
#undef ZR_SYM
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_haloErr;
int FailHalo(const std::string& s) { g_haloErr = s; return -1; }
constexpr int kNcclChar = 0;      // ncclInt8 / ncclChar
}

struct zrh_comm { NcclComm comm = nullptr; int world = 1, rank = 0, device = 0; };

struct zrh_halo_exchange
{
    zr_pass* pass = nullptr; zr_gbuffer* gb = nullptr; zrh_comm* comm = nullptr;
    struct Peer { int rank; size_t sendOff, sendBytes, recvOff, recvBytes; };
    std::vector<Peer> peers;
    std::vector<zr_halo_rect> sendRects, recvRects;
    char* sendBuf = nullptr; char* recvBuf = nullptr; size_t sendBytes = 0, recvBytes = 0;
};

extern "C" {

const char* zrh_halo_last_error(void) { return g_haloErr.c_str(); }

// rank 0 creates the id and hands the 128 bytes to the other ranks by whatever channel launched them (bench.py: torch.distributed broadcast)
int zrh_rccl_unique_id(uint8_t* out128)
{
    std::string err; if (!g_rccl.Load(err)) return FailHalo(err);
    NcclUniqueId id; const int r = g_rccl.GetUniqueId(&id);
    if (r) return FailHalo(std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
    std::memcpy(out128, &id, 128);
    return 0;
}
int zrh_comm_create(int device, int world, int rank, const uint8_t* id128, zrh_comm** out)
{
    std::string err; if (!g_rccl.Load(err)) return FailHalo(err);
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return FailHalo("zrh_comm_create: bad arguments");
    if (hipSetDevice(device) != hipSuccess) return FailHalo("hipSetDevice failed");
    NcclUniqueId id; std::memcpy(&id, id128, 128);
    zrh_comm* c = new zrh_comm(); c->world = world; c->rank = rank; c->device = device;
    const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r) { delete c; return FailHalo(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    *out = c;
    return 0;
}
void zrh_comm_destroy(zrh_comm* c) { if (c) { if (c->comm) g_rccl.CommDestroy(c->comm); delete c; } }

// peers[i]: the rank to talk to and the rects (global pixels; w == 0: nothing in that direction) it needs from us / we need from it --
// tiling.halo_plan's rows.  Both sides derive the same sizes from the tile layout, so there is no handshake.
// bytes_per_pixel: what one exchange of this object moves per pixel; 0 = the pass's own figure (zr_pass_halo_bytes_per_pixel).  A pass with exchanges
// of different sizes (ZR_PASS_DENOISE: ZR_HALO_DENOISE_INPUT 40 B, ZR_HALO_DENOISE_ITER 16 B) gets one object per size on one communicator.
int zrh_halo_exchange_create_bpp(zr_pass* pass, zr_gbuffer* gb, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t n, uint32_t bytes_per_pixel, zrh_halo_exchange** out)
{
    if (!pass || !gb || !comm || !out || (n && !peers)) return FailHalo("zrh_halo_exchange_create: null argument");
    uint32_t bpp = bytes_per_pixel;
    if (!bpp && zr_pass_halo_bytes_per_pixel(pass, &bpp) != ZR_OK) return FailHalo(zr_last_error());
    zrh_halo_exchange* x = new zrh_halo_exchange(); x->pass = pass; x->gb = gb; x->comm = comm;
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    for (uint32_t i = 0; i < n; i++)
    {
        const zrh_halo_peer& p = peers[i];
        if (p.peer < 0 || p.peer >= comm->world) { delete x; return FailHalo("zrh_halo_exchange_create: peer rank out of range"); }
        zrh_halo_exchange::Peer q; q.rank = p.peer; q.sendOff = x->sendBytes; q.recvOff = x->recvBytes;
        q.sendBytes = (size_t)p.send_w * p.send_h * bpp; q.recvBytes = (size_t)p.recv_w * p.recv_h * bpp;
        if (q.sendBytes) { x->sendRects.push_back(zr_halo_rect{p.send_x0, p.send_y0, p.send_w, p.send_h, q.sendOff}); x->sendBytes = align16(x->sendBytes + q.sendBytes); }
        if (q.recvBytes) { x->recvRects.push_back(zr_halo_rect{p.recv_x0, p.recv_y0, p.recv_w, p.recv_h, q.recvOff}); x->recvBytes = align16(x->recvBytes + q.recvBytes); }
        x->peers.push_back(q);
    }
    if (x->sendRects.size() > ZR_HALO_MAX_RECTS || x->recvRects.size() > ZR_HALO_MAX_RECTS) { delete x; return FailHalo("zrh_halo_exchange_create: too many peers"); }
    if (hipSetDevice(comm->device) != hipSuccess || (x->sendBytes && hipMalloc(&x->sendBuf, x->sendBytes) != hipSuccess) ||
        (x->recvBytes && hipMalloc(&x->recvBuf, x->recvBytes) != hipSuccess)) { delete x; return FailHalo("zrh_halo_exchange_create: out of device memory"); }
    *out = x;
    return 0;
}
int zrh_halo_exchange_create(zr_pass* pass, zr_gbuffer* gb, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t n, zrh_halo_exchange** out)
{ return zrh_halo_exchange_create_bpp(pass, gb, comm, peers, n, 0, out); }
void zrh_halo_exchange_destroy(zrh_halo_exchange* x) { if (x) { if (x->sendBuf) (void)hipFree(x->sendBuf); if (x->recvBuf) (void)hipFree(x->recvBuf); delete x; } }
size_t zrh_halo_exchange_send_bytes(const zrh_halo_exchange* x) { return x ? x->sendBytes : 0; }

// pack -> grouped send / recv -> unpack, all enqueued on `stream`; returns without waiting
int zrh_halo_exchange_run(zrh_halo_exchange* x, void* stream, int which)
{
    if (!x) return FailHalo("zrh_halo_exchange_run: null");
    hipStream_t s = (hipStream_t)stream;
    if (x->sendRects.size() && zr_pass_halo_pack_all(x->pass, s, x->gb, which, x->sendRects.data(), (uint32_t)x->sendRects.size(), x->sendBuf, x->sendBytes) != ZR_OK)
        return FailHalo(zr_last_error());
    int r = g_rccl.GroupStart();
    for (const auto& p : x->peers)
    {
        if (!r && p.sendBytes) r = g_rccl.Send(x->sendBuf + p.sendOff, p.sendBytes, kNcclChar, p.rank, x->comm->comm, s);
        if (!r && p.recvBytes) r = g_rccl.Recv(x->recvBuf + p.recvOff, p.recvBytes, kNcclChar, p.rank, x->comm->comm, s);
    }
    const int e = g_rccl.GroupEnd();
    if (r || e) return FailHalo(std::string("RCCL send / recv: ") + g_rccl.GetErrorString(r ? r : e));
    if (x->recvRects.size() && zr_pass_halo_unpack_all(x->pass, s, x->gb, which, x->recvRects.data(), (uint32_t)x->recvRects.size(), x->recvBuf, x->recvBytes) != ZR_OK)
        return FailHalo(zr_last_error());
    return 0;
}

}

// the RenderGraph node (RenderPass-shaped wrapper)
namespace ZetaRayAMD::RenderPass {
void HaloExchange::Init(FrameContext* ctx, zr_pass* pass, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t n, int which)
{
    m_ctx = ctx; m_which = which;
    if (zrh_halo_exchange_create(pass, ctx->gbuffer, comm, peers, n, &m_x) != 0) { std::fprintf(stderr, "HaloExchange: %s\n", zrh_halo_last_error()); std::abort(); }
}
HaloExchange::~HaloExchange() { zrh_halo_exchange_destroy(m_x); }
void HaloExchange::Render(Core::CommandList& cl)
{ if (zrh_halo_exchange_run(m_x, cl.Stream(), m_which) != 0) { std::fprintf(stderr, "HaloExchange: %s\n", zrh_halo_last_error()); std::abort(); } }
}
