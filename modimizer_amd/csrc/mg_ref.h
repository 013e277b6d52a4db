/* mg_ref.h — the device side of modmap's Reference (mg_refpack.hip builds and keeps it, mg_chain.hip's chaining reads it) */
#ifndef MG_REF_H
#define MG_REF_H
#include "mg_common.h"
struct MgRefDev {
  /* what the chaining reads (mg_chain.hip), derived from the arrays of referencePack (modmap.c:74-91) and the modset's flag bytes: */
  U64 *li = 0;                       /* [ms->max + 1] loc[x] | (U64) info[x] << 32: CSR offset and copy class (modmap.c:125-129) of modset index x in one word */
  U64 *revid = 0;                    /* [ref->max + 1] rev[j] | (U64) id[rev[j]] << 32: the j-th occurrence in CSR order with the sequence it lies on; one slot of slack */
  U32 *offset = 0;                   /* per occurrence: position in its sequence (read when an M block is reported) */
  U32 msMax = 0, refMax = 0;         /* what the arrays above were made for */
  int nSeq = 0;
  bool packed = false;               /* they are valid */
  U32 tallies[3] = {0, 0, 0};        /* copy 1 / copy 2 / multiple of the packed reference (modmap.c:125-130): the report of a further file that adds nothing */
  /* while the reference is being read and packed (mg_refpack.hip): */
  U8 *info = 0; U32 *loc = 0, *rev = 0, *id = 0;
  U32 *index = 0, *depth = 0;        /* per occurrence: modset index; per modset index: occurrences */
  size_t capOcc = 0, capMs = 0;
  int device = -1;                   /* the GPU all of this lives on */
};
MgStatus mgRefDevGet (const MgReference *ref, MgRefDev *out);
#endif
