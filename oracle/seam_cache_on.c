/*
 * oracle/seam_cache_on.c -- test infrastructure (see oracle/build_ref.sh).
 *
 * The ONE line a maintainer of the reference would add to its main() to keep
 * the last upload behind the plugin seam,
 *
 *     spmv_seam_cache(2);        // include/hip_csr.h
 *
 * placed in a constructor here so that the reference's main.c is compiled
 * UNMODIFIED into oracle/_ref/ref_dropin_cached as well.  The destructor
 * releases what the cache holds before the library's own teardown.
 */
void spmv_seam_cache(int level);

__attribute__((constructor)) static void seam_cache_on(void) {
    spmv_seam_cache(2);
}

__attribute__((destructor)) static void seam_cache_off(void) {
    spmv_seam_cache(0);
}
