/* mg_ref.h — the device side of modmap's Reference (mg_refpack.hip builds and keeps it, mg_chain.hip's chaining reads it) */
#ifndef MG_REF_H
#define MG_REF_H
#include "mg_common.h"
struct MgRefDev {
  U8 *info = 0;                      /* [ms->max + 1] the modset's flag bytes, copy classes set (modmap.c:125-129) */
  U32 *loc = 0, *rev = 0;            /* referencePack's CSR (modmap.c:74-91): occurrences of index x are rev[loc[x] .. loc[x] + depth[x]) */
  U32 *id = 0, *offset = 0;          /* per occurrence: sequence, position in it */
  U32 msMax = 0, refMax = 0;         /* what the arrays above were made for */
  bool packed = false;               /* loc / rev / info are valid */
  /* while the reference is being read (mgRefBuildAppend): */
  U32 *index = 0, *depth = 0;        /* per occurrence: modset index; per modset index: occurrences */
  size_t capOcc = 0, capMs = 0;
};
MgStatus mgRefDevGet (const MgReference *ref, MgRefDev *out);
#endif
