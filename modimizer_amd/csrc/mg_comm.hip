/* mg_comm.hip — the multi-GPU exchanges of the path, from C, straight on RCCL (BASELINE config 4).
 *
 * Reads shard over the GPUs of a node and every GPU builds its own modset: no collective on the data path.  What IS exchanged:
 *   mgHistogramAllReduce     config 4's global depth histogram: all-reduce (sum) of 65 536 x U64 = 512 KiB per rank over xGMI
 *                            (latency-bound; modutils.c:53-63 is what a rank's histogram is);
 *   mgDepthAllReduce         reads counted against a FIXED modset replicated on every rank (modasm.c:158-174: depth zeroed, then ++depth per
 *                            hit, saturating): the ranks' 16-bit counts widened, all-reduced (sum, uint32) and clamped to 65 535 -- exact,
 *                            a saturating add is associative; 4 bytes per entry per rank, a ring bound by its slowest xGMI link;
 *   mgModsetMergeRankOrder   the exact global modset: the ranks' (value, depth, info) arrays folded into the root's set in RANK
 *                            order with modsetMerge semantics (modset.c:106-128) -- with contiguous blocks of reads per rank this
 *                            reproduces the single-stream build bit for bit (first-occurrence indices, saturated depths); the
 *                            arrays travel point to point (ncclSend / ncclRecv), one rank at a time, from the sender's device
 *                            table into the root's device merge: value[] and depth[] never touch a host.
 * Two ways to get communicators, as RCCL has them: one process driving N devices with one host thread per device
 * (mgCommInitAll; mgSetDevice is per thread), or one process per device (mgCommGetUniqueId on rank 0, the 128 bytes handed to the
 * others by whatever the caller has -- a file, a socket, MPI --, mgCommInitRank everywhere).
 *
 * librccl is loaded when the first communicator is asked for (dlopen): a program that never asks does not map it.
 */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"

struct MgComm { ncclComm_t comm; int rank, size, device; hipStream_t st; U64 *dHist; };

static struct MgRccl
{ void *lib = 0; std::mutex lock;
  ncclResult_t (*GetUniqueId) (ncclUniqueId *) = 0;
  ncclResult_t (*CommInitRank) (ncclComm_t *, int, ncclUniqueId, int) = 0;
  ncclResult_t (*CommInitAll) (ncclComm_t *, int, const int *) = 0;
  ncclResult_t (*CommDestroy) (ncclComm_t) = 0;
  ncclResult_t (*AllReduce) (const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = 0;
  ncclResult_t (*Send) (const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = 0;
  ncclResult_t (*Recv) (void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = 0;
  ncclResult_t (*GroupStart) (void) = 0;
  ncclResult_t (*GroupEnd) (void) = 0;
  const char *(*GetErrorString) (ncclResult_t) = 0;
} gR;

static MgStatus mgRcclLoad (void)
{
  std::lock_guard<std::mutex> g (gR.lock);
  if (gR.lib) return MG_OK;
  void *h = dlopen ("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen ("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen ("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { mgSetError ("librccl not found: %s", dlerror ()); return MG_ERR_NO_DEVICE; }
#define SYM(field, name) do { *(void **) &gR.field = dlsym (h, name); if (!gR.field) { mgSetError ("librccl lacks %s", name); dlclose (h); return MG_ERR_HIP; } } while (0)
  SYM (GetUniqueId, "ncclGetUniqueId"); SYM (CommInitRank, "ncclCommInitRank"); SYM (CommInitAll, "ncclCommInitAll"); SYM (CommDestroy, "ncclCommDestroy");
  SYM (AllReduce, "ncclAllReduce"); SYM (Send, "ncclSend"); SYM (Recv, "ncclRecv"); SYM (GroupStart, "ncclGroupStart"); SYM (GroupEnd, "ncclGroupEnd");
  SYM (GetErrorString, "ncclGetErrorString");
#undef SYM
  gR.lib = h;
  return MG_OK;
}

static MgStatus mgRcclFail (ncclResult_t r, const char *what)
{ mgSetError ("RCCL error %d (%s) in %s", (int) r, gR.GetErrorString ? gR.GetErrorString (r) : "?", what); return MG_ERR_HIP; }
#define MG_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return mgRcclFail (r_, #call); } while (0)

static MgStatus mgCommFinishInit (MgComm *c)      /* the calling thread's device is c->device */
{
  MG_HIP (hipStreamCreateWithFlags (&c->st, hipStreamNonBlocking));
  MG_HIP (hipMalloc ((void **) &c->dHist, 65536 * sizeof (U64)));
  return MG_OK;
}

extern "C" MgStatus mgCommInitAll (MgComm **comms, int nDev, const int *devices)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  if (!comms || nDev < 1 || nDev > mgDeviceCount ()) { mgSetError ("mgCommInitAll: %d devices asked for, %d present", nDev, mgDeviceCount ()); return MG_ERR_ARG; }
  if ((s = mgRcclLoad ())) return s;
  ncclComm_t *cc = (ncclComm_t *) calloc ((size_t) nDev, sizeof (ncclComm_t));
  if (!cc) return MG_ERR_NOMEM;
  ncclResult_t r = gR.CommInitAll (cc, nDev, devices);
  if (r != ncclSuccess) { free (cc); return mgRcclFail (r, "ncclCommInitAll"); }
  int before = 0; (void) hipGetDevice (&before);
  for (int i = 0 ; i < nDev && !s ; ++i)
    { MgComm *c = (MgComm *) calloc (1, sizeof (MgComm));
      if (!c) { s = MG_ERR_NOMEM; break; }
      c->comm = cc[i]; c->rank = i; c->size = nDev; c->device = devices ? devices[i] : i;
      comms[i] = c;
      if (hipSetDevice (c->device) != hipSuccess) { s = mgHipFail (hipGetLastError (), "hipSetDevice"); break; }
      s = mgCommFinishInit (c);
    }
  (void) hipSetDevice (before);
  free (cc);
  return s;
}

extern "C" MgStatus mgCommGetUniqueId (void *id128)
{
  MgStatus s = mgRcclLoad (); if (s) return s;
  ncclUniqueId id; MG_NCCL (gR.GetUniqueId (&id));
  memcpy (id128, &id, sizeof (id) < 128 ? sizeof (id) : 128);
  return MG_OK;
}

extern "C" MgStatus mgCommInitRank (MgComm **comm, int nRanks, int rank, const void *id128, int device)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  if (!comm || nRanks < 1 || rank < 0 || rank >= nRanks || !id128) { mgSetError ("mgCommInitRank: bad arguments"); return MG_ERR_ARG; }
  if ((s = mgRcclLoad ())) return s;
  MG_HIP (hipSetDevice (device));
  ncclUniqueId id; memset (&id, 0, sizeof (id)); memcpy (&id, id128, sizeof (id) < 128 ? sizeof (id) : 128);
  MgComm *c = (MgComm *) calloc (1, sizeof (MgComm));
  if (!c) return MG_ERR_NOMEM;
  ncclResult_t r = gR.CommInitRank (&c->comm, nRanks, id, rank);
  if (r != ncclSuccess) { free (c); return mgRcclFail (r, "ncclCommInitRank"); }
  c->rank = rank; c->size = nRanks; c->device = device;
  if ((s = mgCommFinishInit (c))) { free (c); return s; }
  *comm = c;
  return MG_OK;
}

extern "C" int mgCommRank (const MgComm *c) { return c ? c->rank : -1; }
extern "C" int mgCommSize (const MgComm *c) { return c ? c->size : 0; }

extern "C" void mgCommDestroy (MgComm *c)
{
  if (!c) return;
  int before = 0; (void) hipGetDevice (&before);
  (void) hipSetDevice (c->device);
  if (c->st) { (void) hipStreamSynchronize (c->st); }
  if (c->comm && gR.CommDestroy) (void) gR.CommDestroy (c->comm);
  if (c->st) (void) hipStreamDestroy (c->st);
  (void) hipFree (c->dHist);
  (void) hipSetDevice (before);
  free (c);
}

/* hist[65536] = sum over the ranks of the depth histogram of each rank's ms (host array; every rank gets the sum) */
extern "C" MgStatus mgHistogramAllReduce (Modset *ms, U64 *hist65536, MgComm *c)
{
  if (!ms || !hist65536 || !c) { mgSetError ("mgHistogramAllReduce: null argument"); return MG_ERR_ARG; }
  MG_HIP (hipSetDevice (c->device));
  MG_HIP (hipMemsetAsync (c->dHist, 0, 65536 * sizeof (U64), c->st));
  MgStatus s = modsetDepthHistogramDevice (ms, c->dHist, c->st); if (s) return s;
  MG_NCCL (gR.AllReduce (c->dHist, c->dHist, 65536, ncclUint64, ncclSum, c->comm, c->st));
  MG_HIP (hipMemcpyAsync (hist65536, c->dHist, 65536 * sizeof (U64), hipMemcpyDeviceToHost, c->st));
  MG_HIP (hipStreamSynchronize (c->st));
  return MG_OK;
}

__global__ void mgDepthWidenKernel (const U16 *depth1, U32 n, U32 *wide)
{ for (U32 i = blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += gridDim.x * blockDim.x) wide[i] = depth1[i]; }
__global__ void mgDepthClampKernel (const U32 *wide, U32 n, U16 *depth0)      /* depth0[0] = 0 (entry 0 is nobody's), depth0[i + 1] = min (65535, wide[i]) */
{ for (U32 i = blockIdx.x * blockDim.x + threadIdx.x ; i <= n ; i += gridDim.x * blockDim.x) depth0[i] = i ? (U16) (wide[i - 1] > 0xffffu ? 0xffffu : wide[i - 1]) : (U16) 0; }

/* ms->depth[i] = min (65535, sum over the ranks of their depth[i]), i = 1 .. max, on every rank: in the host array and in the device table.
 * The sets must hold the same entries (one set loaded or built identically everywhere, reads counted per rank): max is compared first. */
extern "C" MgStatus mgDepthAllReduce (Modset *ms, MgComm *c)
{
  if (!ms || !c) { mgSetError ("mgDepthAllReduce: null argument"); return MG_ERR_ARG; }
  MG_HIP (hipSetDevice (c->device));
  MgStatus s = modsetSyncToHost (ms, 0); if (s) return s;      /* ms->depth is the authority (modutils.c:26: callers bump it themselves): counts pending on the device are folded into it first */
  const U32 n = ms->max;
  /* the same set everywhere?  max and its complement, all-reduced with MAX: equal on all ranks iff the two still add up */
  U64 probe[2] = { n, 0xffffffffull - n };
  MG_HIP (hipMemcpy (c->dHist, probe, 16, hipMemcpyHostToDevice));
  MG_NCCL (gR.AllReduce (c->dHist, c->dHist, 2, ncclUint64, ncclMax, c->comm, c->st));
  MG_HIP (hipMemcpyAsync (probe, c->dHist, 16, hipMemcpyDeviceToHost, c->st));
  MG_HIP (hipStreamSynchronize (c->st));
  if (probe[0] + probe[1] != 0xffffffffull) { mgSetError ("mgDepthAllReduce: the ranks' sets differ (this one has %u entries, the largest %llu)", n, (unsigned long long) probe[0]); return MG_ERR_ARG; }
  if (!n) return MG_OK;
  U32 *dWide = 0; U16 *dStage = 0, *dOut = 0;
  s = MG_ERR_HIP;
  do {
    if (hipMalloc ((void **) &dWide, (size_t) n * 4) || hipMalloc ((void **) &dOut, ((size_t) n + 1) * 2) || hipMalloc ((void **) &dStage, (size_t) n * 2) || hipDeviceSynchronize ()) break;
    if ((s = mgXferH2D (dStage, ms->depth + 1, (size_t) n * 2))) break;
    s = MG_ERR_HIP;
    const U16 *dD1 = dStage;
    hipLaunchKernelGGL (mgDepthWidenKernel, dim3 (2048), dim3 (256), 0, c->st, dD1, n, dWide);
    ncclResult_t r = gR.AllReduce (dWide, dWide, n, ncclUint32, ncclSum, c->comm, c->st);
    if (r != ncclSuccess) { mgRcclFail (r, "ncclAllReduce"); break; }
    hipLaunchKernelGGL (mgDepthClampKernel, dim3 (2048), dim3 (256), 0, c->st, dWide, n, dOut);
    if (hipGetLastError () != hipSuccess || hipStreamSynchronize (c->st)) break;
    if ((s = mgXferD2H (ms->depth, dOut, ((size_t) n + 1) * 2, MG_XFER_COPY))) break;
    if ((s = mgModsetAdoptDepthDevice (ms, dOut))) break;      /* the device table's own copy follows (no rebuild from the host on its next use) */
    s = MG_OK;
  } while (0);
  (void) hipFree (dWide); (void) hipFree (dStage); (void) hipFree (dOut);
  if (s == MG_ERR_HIP && !mgLastError ()[0]) mgHipFail (hipGetLastError (), "mgDepthAllReduce");
  return s;
}

/* Every rank calls this.  On `root`, ms afterwards holds the merge of all ranks' sets in rank order (root's own set must be rank
 * 0's for the single-stream identity to hold: root = 0 is the meaningful choice; any root works as "fold the others into mine, in rank
 * order, skipping me").  The other ranks' sets are left as they are. */
extern "C" MgStatus mgModsetMergeRankOrder (Modset *ms, MgComm *c, int root)
{
  if (!ms || !c || root < 0 || root >= c->size) { mgSetError ("mgModsetMergeRankOrder: bad arguments"); return MG_ERR_ARG; }
  MG_HIP (hipSetDevice (c->device));
  MgStatus s = MG_OK;
  U64 *dCount = 0;
  MG_HIP (hipMalloc ((void **) &dCount, 8));
  if (c->rank != root)
    { /* count first, then value / depth / info of entries 1 .. max: value and depth from where the device table keeps them; info lives on
         the host alone (it is the callers' byte) and is staged.  A set without a device table is staged whole. */
      const U64 *sV = 0; const U16 *sD = 0; U32 devMax = 0;
      const bool onDevice = mgHookDeviceView (ms, &sV, &sD, &devMax) == 0;
      if (!onDevice && (s = modsetSyncToHost (ms, 0))) { (void) hipFree (dCount); return s; }
      const U64 n = onDevice ? devMax : ms->max;
      U64 *dV = 0; U16 *dD = 0; U8 *dI = 0;
      do {
        s = MG_ERR_HIP;
        if (hipMalloc ((void **) &dI, n + 1)) break;
        if (!onDevice && (hipMalloc ((void **) &dV, (n + 1) * 8) || hipMalloc ((void **) &dD, (n + 1) * 2))) break;
        if (hipMemcpy (dCount, &n, 8, hipMemcpyHostToDevice) || hipDeviceSynchronize ()) break;
        if (n && mgXferH2D (dI, ms->info + 1, n)) break;
        if (n && !onDevice && (mgXferH2D (dV, ms->value + 1, n * 8) || mgXferH2D (dD, ms->depth + 1, n * 2))) break;
        if (!onDevice) { sV = dV; sD = dD; }
        ncclResult_t r = gR.Send (dCount, 1, ncclUint64, root, c->comm, c->st);
        if (r == ncclSuccess && n)
          { gR.GroupStart ();
            r = gR.Send (sV, n, ncclUint64, root, c->comm, c->st);
            if (r == ncclSuccess) r = gR.Send (sD, n * 2, ncclUint8, root, c->comm, c->st);
            if (r == ncclSuccess) r = gR.Send (dI, n, ncclUint8, root, c->comm, c->st);
            gR.GroupEnd ();
          }
        if (r != ncclSuccess) { mgRcclFail (r, "ncclSend"); break; }
        if (hipStreamSynchronize (c->st)) break;
        s = MG_OK;
      } while (0);
      (void) hipFree (dV); (void) hipFree (dD); (void) hipFree (dI);
    }
  else
    for (int peer = 0 ; peer < c->size && !s ; ++peer)
      { if (peer == root) continue;
        U64 n = 0;
        ncclResult_t r = gR.Recv (dCount, 1, ncclUint64, peer, c->comm, c->st);
        if (r != ncclSuccess) { s = mgRcclFail (r, "ncclRecv"); break; }
        if (hipStreamSynchronize (c->st) || hipMemcpy (&n, dCount, 8, hipMemcpyDeviceToHost)) { s = mgHipFail (hipGetLastError (), "merge recv"); break; }
        if (!n) continue;
        U64 *dV = 0; U16 *dD = 0; U8 *dI = 0;
        U64 *hV = 0; U16 *hD = 0; U8 *hI = 0;
        do {
          s = MG_ERR_HIP;
          if (hipMalloc ((void **) &dV, n * 8) || hipMalloc ((void **) &dD, n * 2) || hipMalloc ((void **) &dI, n)) break;
          gR.GroupStart ();
          r = gR.Recv (dV, n, ncclUint64, peer, c->comm, c->st);
          if (r == ncclSuccess) r = gR.Recv (dD, n * 2, ncclUint8, peer, c->comm, c->st);
          if (r == ncclSuccess) r = gR.Recv (dI, n, ncclUint8, peer, c->comm, c->st);
          gR.GroupEnd ();
          if (r != ncclSuccess) { mgRcclFail (r, "ncclRecv"); break; }
          if (hipStreamSynchronize (c->st)) break;
          if (mgModsetMergeDeviceArrays (ms, dV, dD, dI, (U32) n)) { s = MG_OK; break; }      /* modset.c:106-128 on the device, from where the arrays arrived */
          /* the root's set lives on the host alone: the arrays go there and the host merges */
          hV = (U64 *) mgAllocBig ((n + 1) * 8); hD = (U16 *) mgAllocBig ((n + 1) * 2); hI = (U8 *) mgAllocBig (n + 1);
          if (!hV || !hD || !hI) { s = MG_ERR_NOMEM; break; }
          if (mgXferD2H (hV + 1, dV, n * 8, MG_XFER_COPY) || mgXferD2H (hD + 1, dD, n * 2, MG_XFER_COPY) || mgXferD2H (hI + 1, dI, n, MG_XFER_COPY)) break;
          (void) hipFree (dV); (void) hipFree (dD); (void) hipFree (dI); dV = 0; dD = 0; dI = 0;
          if (!mgModsetMergeArrays (ms, hV, hD, hI, (U32) n)) { mgSetError ("mgModsetMergeRankOrder: merge refused"); s = MG_ERR_ARG; break; }
          s = MG_OK;
        } while (0);
        (void) hipFree (dV); (void) hipFree (dD); (void) hipFree (dI);
        free (hV); free (hD); free (hI);
      }
  (void) hipFree (dCount);
  if (s == MG_ERR_HIP && !mgLastError ()[0]) mgHipFail (hipGetLastError (), "mgModsetMergeRankOrder");
  return s;
}
