/* mg_version.c -- ties the binary to the sources it was built from.
 * The Makefile hashes every source of the library (sha256 over `sha256sum` of the files, first 16 hex digits) into mg_version.inc,
 * rewritten only when the hash changes; modimizer_amd.source_hash () recomputes it from the tree, modimizer_amd.lib () rebuilds a
 * library whose hash differs and __graft_entry__.smoke () asserts equality, so a stale libmodgpu.so cannot pass. */
#include "mg_version.inc"

/* the marker is greppable in the file without loading it (modimizer_amd.binary_hash) */
const char mgSourceHashMarker[] = "MODGPU_SRC_HASH=" MG_SRC_HASH ;

const char *mgSourceHash (void) { return mgSourceHashMarker + 16 ; }
const char *mgVersion (void) { return "modgpu 0.6 (gfx950) src=" MG_SRC_HASH ; }
