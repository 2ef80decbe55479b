/*
 * oracle/ref_harness.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Builds the *real* reference (rsjahnige/CRYSTALS-Kyber) into oracle/_ref/libmlkem_ref.so
 * straight from the sources where they lie under $MLKEM_REF_DIR (default /root/reference).
 * Nothing of the reference is copied: this translation unit textually includes
 * <ref>/ml_kem.c so that its `static` deterministic entry points
 * (KeyGen_internal ml_kem.c:1034, Encaps_internal :1093, Decaps_internal :1136, PKE_* :651/:776/:942,
 * MultiplyNTTs :415, Compress :83, ByteEncode :125 ...) become callable, and wraps each of
 * them behind a packed-uint8 / uint16 C ABI.  <ref>/sha3.c is compiled as a second TU by
 * oracle/Makefile.
 *
 * The reference stores every "byte" in a 4-byte `union byte`; the wrappers
 * widen/narrow at the edge so that callers (tests, golden generator, bench cpu_baseline)
 * only ever see packed bytes.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.
 */
#include "ml_kem.c" /* resolved through -I$(MLKEM_REF_DIR); see oracle/Makefile */

#include <stdint.h>
#include <string.h>

static union byte *widen(const uint8_t *src, size_t n) {
    union byte *w = malloc(sizeof(union byte) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) { w[i].e = 0; w[i].e = src[i]; }
    return w;
}
static void narrow(uint8_t *dst, const union byte *src, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (uint8_t)src[i].e;
}
static union integer *widen_poly(const uint16_t *f) {
    union integer *w = malloc(sizeof(union integer) * N);
    for (int i = 0; i < N; i++) { w[i].l = 0; w[i].t = f[i]; }
    return w;
}
static void narrow_poly(uint16_t *dst, const union integer *src) {
    for (int i = 0; i < N; i++) dst[i] = (uint16_t)src[i].t;
}

int ref_sizes(int set, unsigned *ek_len, unsigned *dk_len, unsigned *c_len) {
    ml_errno = 0;
    struct PARAMS p = init((enum ML_KEM)set);
    if (ml_errno) return ml_errno;
    *ek_len = 384 * p.k.e + 32;
    *dk_len = 768 * p.k.e + 96;
    *c_len = 32 * (p.du.e * p.k.e + p.dv.e);
    return 0;
}

/* ---- ML-KEM internal (deterministic) algorithms ------------------------------------ */

int ref_keygen(int set, const uint8_t d[32], const uint8_t z[32], uint8_t *ek, uint8_t *dk) {
    struct PARAMS p = init((enum ML_KEM)set);
    union byte *wd = widen(d, 32), *wz = widen(z, 32);
    struct PKE keys = KeyGen_internal(&p, wd, wz);
    narrow(ek, keys.ek, keys.ek_len);
    narrow(dk, keys.dk, keys.dk_len);
    free(keys.ek); free(keys.dk); free(wd); free(wz);
    return 0;
}

int ref_encaps(int set, const uint8_t *ek, const uint8_t m[32], uint8_t *c, uint8_t K[32]) {
    struct PARAMS p = init((enum ML_KEM)set);
    union byte *wek = widen(ek, 384 * p.k.e + 32), *wm = widen(m, 32);
    struct KEM kc = Encaps_internal(&p, wek, wm);
    narrow(c, kc.c, kc.c_len);
    narrow(K, kc.K, 32);
    free(kc.c); free(wek); free(wm);
    return 0;
}

/* Decaps_internal: no input checking (ml_kem.c:1136) */
int ref_decaps_internal(int set, const uint8_t *dk, const uint8_t *c, uint8_t K[32]) {
    struct PARAMS p = init((enum ML_KEM)set);
    unsigned c_len = 32 * (p.du.e * p.k.e + p.dv.e);
    union byte *wdk = widen(dk, 768 * p.k.e + 96), *wc = widen(c, c_len);
    union byte *k = Decaps_internal(&p, wdk, wc);
    narrow(K, k, 32);
    free(k); free(wdk); free(wc);
    return 0;
}

/* Public KEM_Decaps incl. length + hash checks (ml_kem.c:1310); returns ml_errno (0 on success). */
int ref_kem_decaps(int set, const uint8_t *dk, unsigned dk_len, const uint8_t *c, unsigned c_len,
                   uint8_t K[32]) {
    struct PARAMS p = init((enum ML_KEM)set);
    union byte *wdk = widen(dk, dk_len), *wc = widen(c, c_len);
    ml_errno = 0;
    union byte *k = KEM_Decaps(&p, wdk, dk_len, wc, c_len);
    int rc = ml_errno;
    if (k) { narrow(K, k, 32); free(k); }
    free(wdk); free(wc);
    ml_errno = 0;
    return rc;
}

/* Public KEM_Encaps input checks only (ml_kem.c:1257-1291): returns ml_errno; K/c are random
 * (urandom-seeded) and therefore only useful for round-trip checks. */
int ref_kem_encaps_public(int set, const uint8_t *ek, unsigned ek_len, uint8_t *c, uint8_t K[32]) {
    struct PARAMS p = init((enum ML_KEM)set);
    union byte *wek = widen(ek, ek_len);
    ml_errno = 0;
    struct KEM kc;
    memset(&kc, 0, sizeof kc);
    kc = KEM_Encaps(&p, wek, ek_len);
    int rc = ml_errno;
    if (rc == 0) {
        narrow(c, kc.c, kc.c_len);
        narrow(K, kc.K, 32);
        free(kc.c);
    }
    free(wek);
    ml_errno = 0;
    return rc;
}

int ref_kem_keygen_public(int set, uint8_t *ek, uint8_t *dk) {
    struct PARAMS p = init((enum ML_KEM)set);
    ml_errno = 0;
    struct PKE keys = KEM_KeyGen(&p);
    int rc = ml_errno;
    if (rc == 0) {
        narrow(ek, keys.ek, keys.ek_len);
        narrow(dk, keys.dk, keys.dk_len);
        free(keys.ek); free(keys.dk);
    }
    ml_errno = 0;
    return rc;
}

int ref_init_errno(int set) {
    ml_errno = 0;
    (void)init((enum ML_KEM)set);
    int rc = ml_errno;
    ml_errno = 0;
    return rc;
}

/* ---- K-PKE ------------------------------------------------------------------------- */

int ref_pke_keygen(int set, const uint8_t d[32], uint8_t *ek, uint8_t *dk_pke) {
    struct PARAMS p = init((enum ML_KEM)set);
    union byte *wd = widen(d, 32);
    struct PKE keys = PKE_KeyGen(&p, wd);
    narrow(ek, keys.ek, keys.ek_len);
    narrow(dk_pke, keys.dk, keys.dk_len);
    free(keys.ek); free(keys.dk); free(wd);
    return 0;
}

int ref_pke_encrypt(int set, const uint8_t *ek, const uint8_t m[32], const uint8_t r[32], uint8_t *c) {
    struct PARAMS p = init((enum ML_KEM)set);
    unsigned c_len = 32 * (p.du.e * p.k.e + p.dv.e);
    union byte *wek = widen(ek, 384 * p.k.e + 32), *wm = widen(m, 32), *wr = widen(r, 32);
    union byte *wc = PKE_Encrypt(&p, wek, wm, wr);
    narrow(c, wc, c_len);
    free(wc); free(wek); free(wm); free(wr);
    return 0;
}

int ref_pke_decrypt(int set, const uint8_t *dk_pke, const uint8_t *c, uint8_t m[32]) {
    struct PARAMS p = init((enum ML_KEM)set);
    unsigned c_len = 32 * (p.du.e * p.k.e + p.dv.e);
    union byte *wdk = widen(dk_pke, 384 * p.k.e), *wc = widen(c, c_len);
    union byte *wm = PKE_Decrypt(&p, wdk, wc);
    narrow(m, wm, 32);
    free(wm); free(wdk); free(wc);
    return 0;
}

/* ---- polynomial primitives ---------------------------------------------------------- */

void ref_ntt(const uint16_t f[256], uint16_t fh[256]) {
    union integer *w = widen_poly(f), *r = NTT(w);
    narrow_poly(fh, r); free(w); free(r);
}
void ref_intt(const uint16_t fh[256], uint16_t f[256]) {
    union integer *w = widen_poly(fh), *r = InverseNTT(w);
    narrow_poly(f, r); free(w); free(r);
}
void ref_multiply_ntts(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]) {
    union integer *wa = widen_poly(a), *wb = widen_poly(b), *r = MultiplyNTTs(wa, wb);
    narrow_poly(h, r); free(wa); free(wb); free(r);
}
void ref_poly_add(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]) {
    union integer *wa = widen_poly(a), *wb = widen_poly(b), *r = PolyAddition(wa, wb);
    narrow_poly(h, r); free(wa); free(wb); free(r);
}
void ref_poly_sub(const uint16_t a[256], const uint16_t b[256], uint16_t h[256]) {
    union integer *wa = widen_poly(a), *wb = widen_poly(b), *r = PolySubtraction(wa, wb);
    narrow_poly(h, r); free(wa); free(wb); free(r);
}
/* SampleNTT mutates its argument on the (unreachable) retry path, so hand it a private copy. */
void ref_sample_ntt(const uint8_t B[34], uint16_t a[256]) {
    union byte *w = widen(B, 34);
    union integer *r = SampleNTT(w);
    narrow_poly(a, r); free(w); free(r);
}
void ref_sample_cbd(const uint8_t *B, unsigned eta, uint16_t f[256]) {
    union byte *w = widen(B, 64 * eta);
    union integer *r = SamplePolyCBD(w, eta);
    narrow_poly(f, r); free(w); free(r);
}
unsigned ref_compress(unsigned x, unsigned d) {
    union integer v; v.l = 0; v.t = x;
    return Compress(v, d).t;
}
unsigned ref_decompress(unsigned y, unsigned d) {
    union integer v; v.l = 0; v.t = y;
    return Decompress(v, d).t;
}
void ref_byte_encode(const uint16_t F[256], unsigned d, uint8_t *B) {
    union integer *w = widen_poly(F);
    union byte *r = ByteEncode(w, d);
    narrow(B, r, 32 * d); free(w); free(r);
}
void ref_byte_decode(const uint8_t *B, unsigned d, uint16_t F[256]) {
    union byte *w = widen(B, 32 * d);
    union integer *r = ByteDecode(w, d);
    narrow_poly(F, r); free(w); free(r);
}
unsigned ref_bitrev7(unsigned r) {
    union byte b; b.e = 0; b.s = r;
    return BitRev7(b).s;
}

/* ---- hash wrappers (PRF and J are SHAKE128 in the reference: SURVEY.md F2) ----------- */

void ref_prf(const uint8_t s[32], uint8_t b, unsigned eta, uint8_t *out) {
    union byte *ws = widen(s, 32), wb; wb.e = b;
    union byte *r = PRF(ws, wb, eta);
    narrow(out, r, 64 * eta); free(ws); free(r);
}
void ref_H(const uint8_t *in, unsigned len, uint8_t out[32]) {
    union byte *w = widen(in, len), *r = H(w, len);
    narrow(out, r, 32); free(w); free(r);
}
void ref_J(const uint8_t *in, unsigned len, uint8_t out[32]) {
    union byte *w = widen(in, len), *r = J(w, len);
    narrow(out, r, 32); free(w); free(r);
}
void ref_G(const uint8_t *in, unsigned len, uint8_t out[64]) {
    union byte *w = widen(in, len), *r = G(w, len);
    narrow(out, r, 64); free(w); free(r);
}

/* Bit-granular sponge entry (sha3.c:408): msg_bits[i] in {0,1}; out_bits receives d bits.
 * suffix_kind: 0 = hash "01", 1 = XOF "1111". */
void ref_sha3_bits(const uint8_t *msg_bits, unsigned n, unsigned d, unsigned c, int suffix_kind,
                   uint8_t *out_bits) {
    union bit *in = malloc(sizeof(union bit) * (n ? n : 1));
    for (unsigned i = 0; i < n; i++) in[i].b = msg_bits[i] & 1;
    union bit sfx_hash[4] = {{0}, {1}, {0}, {0}};
    union bit sfx_xof[4] = {{1}, {1}, {1}, {1}};
    union bit *r = sha3_b(in, n, d, c, suffix_kind ? sfx_xof : sfx_hash);
    for (unsigned i = 0; i < d; i++) out_bits[i] = (uint8_t)r[i].b;
    free(in); free(r);
}

/* The same with the caller's suffix cells handed to sha3_b verbatim (sha3.c:414-429): sfx[2] == 1 selects four suffix
 * bits, otherwise sfx[0], sfx[1] are appended ("11" = RawSHAKE). */
void ref_sha3_bits_sfx(const uint8_t *msg_bits, unsigned n, unsigned d, unsigned c, const uint8_t sfx4[4],
                       uint8_t *out_bits) {
    union bit *in = malloc(sizeof(union bit) * (n ? n : 1));
    for (unsigned i = 0; i < n; i++) in[i].b = msg_bits[i] & 1;
    union bit sfx[4];
    for (int i = 0; i < 4; i++) sfx[i].b = sfx4[i] & 1;
    union bit *r = sha3_b(in, n, d, c, sfx);
    for (unsigned i = 0; i < d; i++) out_bits[i] = (uint8_t)r[i].b;
    free(in); free(r);
}

/* ---- timing helper for bench.py's cpu_baseline ("reference" kind) -------------------- */
#include <time.h>
/* Runs Encaps_internal + KEM_Decaps on `pairs` key/message sets; returns elapsed seconds and
 * the number of pairs whose two shared secrets agree. */
double ref_time_encaps_decaps(int set, int pairs, const uint8_t *ek, const uint8_t *dk,
                              const uint8_t *m, uint8_t *c_out, uint8_t *K_out, int *agree) {
    unsigned ek_len, dk_len, c_len;
    ref_sizes(set, &ek_len, &dk_len, &c_len);
    struct timespec t0, t1;
    int ok = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < pairs; i++) {
        uint8_t K2[32];
        ref_encaps(set, ek + (size_t)i * ek_len, m + 32 * (size_t)i, c_out + (size_t)i * c_len,
                   K_out + 32 * (size_t)i);
        int rc = ref_kem_decaps(set, dk + (size_t)i * dk_len, dk_len, c_out + (size_t)i * c_len, c_len, K2);
        if (rc == 0 && memcmp(K2, K_out + 32 * (size_t)i, 32) == 0) ok++;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *agree = ok;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* KeyGen_internal + Encaps_internal + KEM_Decaps on `n` seed sets (BASELINE configs[3] as a CPU leg); returns elapsed
 * seconds, the keys / ciphertexts / shared secrets, and the number of items whose two shared secrets agree. */
double ref_time_triples(int set, int n, const uint8_t *d, const uint8_t *z, const uint8_t *m, uint8_t *ek_out,
                        uint8_t *dk_out, uint8_t *c_out, uint8_t *K_out, int *agree) {
    unsigned ek_len, dk_len, c_len;
    ref_sizes(set, &ek_len, &dk_len, &c_len);
    struct timespec t0, t1;
    int ok = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < n; i++) {
        uint8_t K2[32];
        uint8_t *ek = ek_out + (size_t)i * ek_len, *dk = dk_out + (size_t)i * dk_len, *c = c_out + (size_t)i * c_len;
        ref_keygen(set, d + 32 * (size_t)i, z + 32 * (size_t)i, ek, dk);
        ref_encaps(set, ek, m + 32 * (size_t)i, c, K_out + 32 * (size_t)i);
        int rc = ref_kem_decaps(set, dk, dk_len, c, c_len, K2);
        if (rc == 0 && memcmp(K2, K_out + 32 * (size_t)i, 32) == 0) ok++;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *agree = ok;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
