/* sanitize_main.c -- runs every entry point of the CPU restatement (vrt_oracle.c) under AddressSanitizer +
 * UndefinedBehaviorSanitizer on a small scene.  TEST INFRASTRUCTURE ONLY (tests/test_oracle_sanitize.py builds and runs
 * it; the reference's own `make test` runs its unit tests under valgrind, Makefile:99-101 -- sanitizers are this
 * repository's equivalent for the CPU-side checker; GPU sanitizers are not available on the MI355X pool).
 *
 *   sanitize_main <dir> N nrays iters
 * reads  <dir>/{ior.f32, ior.u32, tr.u32, sp.u32, sd.f32, sd.i16}   (volume N^3, nrays rays in VOLUME coordinates)
 * writes <dir>/{ep_f32.u32, ed_f32.f32, ei_f32.u32, ep_i16.u32, ed_i16.i16, ei_i16.u32}: the scene-level results of both
 * families (scene build -> interleave -> ray prologue -> march), which the test compares with the normal build's bits.
 * Every buffer is heap-allocated at its exact size so that an out-of-bounds access is seen.
 */
#include "vrt_oracle.c"

#include <stdio.h>

static void *load(const char *dir, const char *name, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void *p = malloc(bytes ? bytes : 1);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return p;
}

static void save(const char *dir, const char *name, const void *p, size_t bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s dir N nrays iters\n", argv[0]); return 2; }
    const char *dir = argv[1];
    const size_t N = (size_t)atol(argv[2]), nrays = (size_t)atol(argv[3]);
    const uint32_t iters = (uint32_t)atol(argv[4]);
    const size_t nvox = N * N * N, O = N - 2, nf = O * O * O;
    const size_t vb[3] = {N, N, N};
    const uint16_t fb[3] = {(uint16_t)O, (uint16_t)O, (uint16_t)O};
    const float inv[3] = {1.0f, 1.0f, 1.0f};
    uint32_t *tr = load(dir, "tr.u32", nvox * 4);
    uint32_t *sp = load(dir, "sp.u32", nrays * 12);
    unsigned long long sum = 0;

    for (int fam = 0; fam < 2; ++fam) {
        const size_t esz = fam == 0 ? 4 : 2;            /* field / direction element size */
        void *ior = load(dir, fam == 0 ? "ior.f32" : "ior.u32", nvox * 4);
        void *sd = load(dir, fam == 0 ? "sd.f32" : "sd.i16", nrays * 3 * esz);
        void *dx = malloc(nf * esz), *dy = malloc(nf * esz), *dz = malloc(nf * esz);
        uint32_t *tc = malloc(nf * 4);
        int rc = fam == 0 ? vrt_oracle_scene_build_f32(vb, ior, tr, dx, dy, dz, tc)
                          : vrt_oracle_scene_build_i16(vb, ior, tr, dx, dy, dz, tc);
        if (rc) { fprintf(stderr, "scene build rc %d\n", rc); return 3; }
        void *field = malloc(nf * 4 * esz);
        if (fam == 0) vrt_oracle_interleave_f32(dx, dy, dz, tc, nf, field);
        else          vrt_oracle_interleave_i16(dx, dy, dz, tc, nf, field);
        uint32_t *pos = malloc(nrays * 12);
        memcpy(pos, sp, nrays * 12);
        const long prc = vrt_oracle_ray_prologue(fam == 0 ? VRT_F32 : VRT_I16, vb, ior, nrays, pos, fam == 0 ? VRT_F32 : VRT_I16, sd);
        if (prc) { fprintf(stderr, "prologue rc %ld\n", prc); return 3; }
        uint32_t *ep = malloc(nrays * 12), *ei = malloc(nrays * 4), *rl = malloc(nrays * 4);
        void *ed = malloc(nrays * 3 * esz);
        const int ft = fam == 0 ? VRT_F32 : VRT_I16;
        rc = vrt_oracle_trace(ft, ft, field, fb, inv, nrays, pos, sd, ep, ed, ei, rl, NULL, iters, 0);
        if (rc) { fprintf(stderr, "trace rc %d\n", rc); return 3; }
        save(dir, fam == 0 ? "ep_f32.u32" : "ep_i16.u32", ep, nrays * 12);
        save(dir, fam == 0 ? "ed_f32.f32" : "ed_i16.i16", ed, nrays * 3 * esz);
        save(dir, fam == 0 ? "ei_f32.u32" : "ei_i16.u32", ei, nrays * 4);
        for (size_t i = 0; i < nrays; ++i) sum += ei[i];
        /* path tracing (exactly iterations * 3 * nrays words) and the opt-in absorption branch, on a few rays */
        const size_t np = nrays < 64 ? nrays : 64;
        const uint32_t pit = 97;
        uint32_t *path = malloc((size_t)pit * 3 * np * 4);
        rc = vrt_oracle_trace(ft, ft, field, fb, inv, np, pos, sd, ep, ed, ei, rl, path, pit, 1);
        if (rc) { fprintf(stderr, "path trace rc %d\n", rc); return 3; }
        for (size_t i = 0; i < (size_t)pit * 3 * np; ++i) sum += path[i] & 0xFFu;
        rc = vrt_oracle_trace_absorb(ft, ft, field, fb, inv, nrays, pos, sd, ep, ed, ei, rl, NULL, iters, 2, tc, 1000u);
        if (rc) { fprintf(stderr, "absorb trace rc %d\n", rc); return 3; }
        for (size_t i = 0; i < nrays; ++i) sum += rl[i] & 0xFFu;
        free(path); free(ep); free(ed); free(ei); free(rl); free(pos); free(field);
        free(dx); free(dy); free(dz); free(tc); free(ior); free(sd);
    }
    free(tr); free(sp);
    printf("sanitize_main ok, checksum %llu, threads %d\n", sum, vrt_oracle_num_threads());
    return 0;
}
