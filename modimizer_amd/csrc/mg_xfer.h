/* mg_xfer.h — whole arrays between HBM and pageable host arrays, by a team of threads through page-locked blocks (mg_xfer.hip) */
#ifndef MG_XFER_H
#define MG_XFER_H
#include "modgpu.h"
#define MG_XFER_COPY     0      /* dst[i] = src[i] */
#define MG_XFER_SATADD16 1      /* U16: dst[i] = min (65535, dst[i] + src[i])   (modutils.c:26) */
/* both return when the bytes are where they go; the caller has synchronised whatever produced devSrc / reads devDst */
MgStatus mgXferD2H (void *hostDst, const void *devSrc, size_t bytes, int op);
MgStatus mgXferH2D (void *devDst, const void *hostSrc, size_t bytes);
MgStatus mgXferH2DSparse (void *devDst, const void *hostSrc, size_t bytes);      /* the same for a calloc ()ed array of the library's that may be mostly untouched: never-written pages are not read (see mg_xfer.hip) */
int      mgXferThreads (void);
void     mgXferWarm (void);            /* make the team's streams and page-locked blocks now, on a thread of its own (a transfer is going to follow) */
#ifdef __cplusplus
extern "C" {
#endif
void     mgXferReleaseBuffers (void);
#ifdef __cplusplus
}
#endif
#endif
