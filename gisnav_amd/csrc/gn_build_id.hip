// The digest of the sources and compile flags this library was built from (gisnav_amd/build.py passes it on the command line and recompiles this
// file whenever it changes).  gisnav_amd._lib.load compares it with the digest of the tree it runs from and refuses a stale binary.
#ifndef GN_SOURCE_DIGEST
#error "build through gisnav_amd/build.py: it defines GN_SOURCE_DIGEST"
#endif
extern "C" const char* gn_source_digest(void) { return GN_SOURCE_DIGEST; }
extern "C" const char* gn_version(void) { return "gisnav_amd 0.3.0 gfx950 src:" GN_SOURCE_DIGEST; }
