/*
 * ml_kem_shim.c — libml_kem.so: the reference's public C API (ml_kem.h) on top of the MI355X batch engine.
 *
 * Each call widens/narrows between the reference's 4-byte `union byte` cells and the packed bytes of the
 * batch C-ABI (include/mlkem_batch.h) and runs a batch of one on the GPU.  There is no CPU fallback: if no
 * HIP device is usable the call prints an error, sets ml_errno = -100 and returns a zeroed result.
 * Error codes, messages and ownership follow ml_kem.c (init :1363, KEM_KeyGen :1233, KEM_Encaps :1257,
 * KEM_Decaps :1310); unlike the reference, error paths return zeroed structs instead of uninitialised ones.
 */
#define _DEFAULT_SOURCE   /* explicit_bzero */
#include "../../include/mlkem_compat.h"
#include "../../include/mlkem_batch.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int ml_errno = 0;

static void report(const char* where, const char* msg) {
    fprintf(stderr, "ERROR: %s\n", where);
    fprintf(stderr, "\t%s\n", msg);
}

static int set_of(const struct PARAMS* p) {
    switch (p->k.e) {
    case 2: return 512;
    case 3: return 768;
    case 4: return 1024;
    default: return 0;
    }
}

static union byte* widen(const unsigned char* src, size_t n) {
    union byte* w = (union byte*)calloc(n ? n : 1, sizeof(union byte));
    if (w) for (size_t i = 0; i < n; i++) w[i].e = src[i];
    return w;
}
static unsigned char* narrow(const union byte* src, size_t n) {
    unsigned char* b = (unsigned char*)malloc(n ? n : 1);
    if (b) for (size_t i = 0; i < n; i++) b[i] = (unsigned char)src[i].e;   /* upper 24 bits are never read */
    return b;
}
/* packed staging copies of keys and shared secrets are zeroed before they are released */
static void wipe_free(unsigned char* p, size_t n) {
    if (!p) return;
    explicit_bzero(p, n);
    free(p);
}
static int engine_failed(const char* where, int rc) {
    if (rc == 0) return 0;
    report(where, mlkem_strerror(rc));
    if (rc == MLKEM_ERR_NO_DEVICE) fprintf(stderr, "\t%s\n", mlkem_last_hip_error());
    ml_errno = rc;
    return 1;
}

const struct PARAMS init(enum ML_KEM param_set) {
    struct PARAMS params;
    int v[5];
    memset(&params, 0, sizeof params);
    if (mlkem_params((int)param_set, v) != 0) {
        report("ml_kem shim - init()", "init() :: Invalid paramater set provided\n");
        ml_errno = -1;
        return params;
    }
    params.k.e = v[0]; params.n1.e = v[1]; params.n2.e = v[2]; params.du.e = v[3]; params.dv.e = v[4];
    return params;
}

struct PKE KEM_KeyGen(const struct PARAMS* params) {
    struct PKE result;
    unsigned ek_len, dk_len, c_len;
    memset(&result, 0, sizeof result);
    const int set = set_of(params);
    if (mlkem_sizes(set, &ek_len, &dk_len, &c_len) != 0) { ml_errno = -1; return result; }
    unsigned char* ek = (unsigned char*)malloc(ek_len);
    unsigned char* dk = (unsigned char*)malloc(dk_len);
    int rc = (ek && dk) ? mlkem_keygen_random(set, 1, ek, dk) : MLKEM_ERR_ALLOC;
    if (rc == MLKEM_ERR_RNG) {
        report("ml_kem shim - KEM_KeyGen()", "KEM_KeyGen() :: Random bit generation failed\n");
        ml_errno = -2;
    } else if (!engine_failed("ml_kem shim - KEM_KeyGen()", rc)) {
        result.ek = widen(ek, ek_len);
        result.dk = widen(dk, dk_len);
        result.ek_len = ek_len;
        result.dk_len = dk_len;
    }
    free(ek); wipe_free(dk, dk_len);
    return result;
}

struct KEM KEM_Encaps(const struct PARAMS* params, const union byte* ek, unsigned int ek_len) {
    struct KEM result;
    unsigned want_ek, dk_len, c_len;
    memset(&result, 0, sizeof result);
    const int set = set_of(params);
    if (mlkem_sizes(set, &want_ek, &dk_len, &c_len) != 0) { ml_errno = -1; return result; }
    if (want_ek != ek_len) {   /* ml_kem.c:1267-1271 */
        report("ml_kem shim - KEM_Encaps()", "KEM_Encaps() :: Type check failed\n");
        ml_errno = -3;
        return result;
    }
    /* ml_kem.c:1273-1291: the reference's modulus check cannot fail (ByteDecode_12 never reduces) */
    unsigned char* pek = narrow(ek, ek_len);
    unsigned char* c = (unsigned char*)malloc(c_len);
    unsigned char K[32];
    int rc = (pek && c) ? mlkem_encaps_random(set, 1, pek, ek_len, c, K) : MLKEM_ERR_ALLOC;
    if (rc == MLKEM_ERR_RNG) {
        report("ml_kem shim - KEM_Encaps()", "KEM_Encaps() :: Random bit generation failed\n");
        ml_errno = -2;
    } else if (!engine_failed("ml_kem shim - KEM_Encaps()", rc)) {
        for (int i = 0; i < 32; i++) result.K[i].e = K[i];
        result.c = widen(c, c_len);
        result.c_len = c_len;
    }
    free(pek); free(c);
    explicit_bzero(K, sizeof K);
    return result;
}

union byte* KEM_Decaps(const struct PARAMS* params, const union byte* dk, unsigned int dk_len, const union byte* c,
                       unsigned int c_len) {
    unsigned ek_len, want_dk, want_c;
    const int set = set_of(params);
    if (mlkem_sizes(set, &ek_len, &want_dk, &want_c) != 0) { ml_errno = -1; return NULL; }
    if (c_len != want_c) {     /* ml_kem.c:1320-1325 */
        report("ml_kem shim - KEM_Decaps()", "KEM_Decaps() :: Ciphertext type check failed\n");
        ml_errno = -3;
        return NULL;
    }
    if (dk_len != want_dk) {   /* ml_kem.c:1328-1333 */
        report("ml_kem shim - KEM_Decaps()", "KEM_Decaps() :: Decapsulation key type check failed\n");
        ml_errno = -3;
        return NULL;
    }
    unsigned char* pdk = narrow(dk, dk_len);
    unsigned char* pc = narrow(c, c_len);
    unsigned char K[32];
    int status = 0;
    union byte* result = NULL;
    int rc = (pdk && pc) ? mlkem_decaps(set, 1, pdk, pc, K, &status) : MLKEM_ERR_ALLOC;
    if (!engine_failed("ml_kem shim - KEM_Decaps()", rc)) {
        if (status == MLKEM_ERR_HASH) {   /* ml_kem.c:1336-1350 */
            report("ml_kem shim - KEM_Decaps()", "KEM_Decaps() :: Hash check failed\n");
            ml_errno = -5;
        } else {
            result = widen(K, 32);
        }
    }
    wipe_free(pdk, dk_len); free(pc);
    explicit_bzero(K, sizeof K);
    return result;
}

/* ------------------------------------------------------------------------------------------------------------
 * The four primitives ml_kem.o exports without declaring them (ml_kem.c:189, :253, :287, :336): 4-byte
 * `union integer` cells in and out, result malloc()ed for the caller, a batch of one on the GPU.  On failure they
 * print a message, set ml_errno and return NULL (the reference cannot fail here).
 * ---------------------------------------------------------------------------------------------------------- */
static union integer* widen_poly(const unsigned short* c) {
    union integer* f = (union integer*)calloc(256, sizeof(union integer));
    if (f) for (int i = 0; i < 256; i++) f[i].l = c[i];   /* t = low 12 bits of l: both fields read back the value */
    return f;
}
static void narrow_poly(const union integer* f, unsigned short* c) {
    for (int i = 0; i < 256; i++) c[i] = (unsigned short)f[i].t;
}
union integer* SampleNTT(union byte* B) {   /* ml_kem.c:189-245 */
    unsigned char seed[34], retries = 0;
    unsigned short a[256];
    if (!B) return NULL;
    for (int i = 0; i < 34; i++) seed[i] = (unsigned char)B[i].e;
    if (engine_failed("SampleNTT()", mlkem_sample_ntt_retries(1, seed, a, &retries))) return NULL;
    /* ml_kem.c:237-242: on exhausting its triples the reference does B[32].e += 1, B[33].e += 1 in the CALLER's buffer and
     * starts over; the engine reports how often (0 for every real SHAKE stream) and the same 8-bit fields are advanced here */
    B[32].e += retries;
    B[33].e += retries;
    return widen_poly(a);
}
union integer* SamplePolyCBD(const union byte* B, unsigned int n) {   /* ml_kem.c:253-275 */
    unsigned char bytes[192];
    unsigned short f[256];
    if (!B || (n != 2 && n != 3)) {
        report("SamplePolyCBD()", "eta must be 2 or 3");
        ml_errno = MLKEM_ERR_ARG;
        return NULL;
    }
    for (unsigned i = 0; i < 64 * n; i++) bytes[i] = (unsigned char)B[i].e;
    if (engine_failed("SamplePolyCBD()", mlkem_sample_cbd((int)n, 1, bytes, f))) return NULL;
    return widen_poly(f);
}
union integer* NTT(const union integer* f) {   /* ml_kem.c:287-329 */
    unsigned short in[256], out[256];
    if (!f) return NULL;
    narrow_poly(f, in);
    if (engine_failed("NTT()", mlkem_ntt(1, in, out))) return NULL;
    return widen_poly(out);
}
union integer* InverseNTT(const union integer* fh) {   /* ml_kem.c:336-384 */
    unsigned short in[256], out[256];
    if (!fh) return NULL;
    narrow_poly(fh, in);
    if (engine_failed("InverseNTT()", mlkem_intt(1, in, out))) return NULL;
    return widen_poly(out);
}

/* ------------------------------------------------------------------------------------------------------------
 * sha3.h front-ends: same signatures, cell types and ownership as sha3.c:329-494.  Bit fiddling
 * (hex <-> bits, suffix, pad10*1) happens here on the host; the sponge itself runs on the GPU through
 * mlkem_keccak_sponge (a batch of one).  Failures (unsupported capacity, no device) print a message and return NULL.
 * ---------------------------------------------------------------------------------------------------------- */
union bit* h2b(const union hex* H, unsigned int m, unsigned int n) {   /* sha3.c:329-359 */
    unsigned int total = 8 * m, keep = n < total ? n : total;
    union bit* S = (union bit*)calloc(keep ? keep : 1, sizeof(union bit));
    if (!S) return NULL;
    for (unsigned int i = 0; i < keep; i++) {
        unsigned int byte = 16 * H[2 * (i / 8)].d + H[2 * (i / 8) + 1].d;
        S[i].b = (byte >> (i % 8)) & 1u;
    }
    return S;
}

union hex* b2h(const union bit* S, unsigned int n) {   /* sha3.c:367-396 */
    unsigned int m = (n + 7) / 8;
    union hex* H = (union hex*)calloc(m ? 2 * (size_t)m : 1, sizeof(union hex));
    if (!H) return NULL;
    for (unsigned int i = 0; i < m; i++) {
        unsigned int byte = 0;
        for (unsigned int j = 0; j < 8; j++)
            if (8 * i + j < n) byte |= (S[8 * i + j].b & 1u) << j;
        H[2 * i].d = (byte >> 4) & 15u;
        H[2 * i + 1].d = byte & 15u;
    }
    return H;
}

union bit* sha3_b(const union bit* bstr, unsigned int n, unsigned int d, unsigned int c, union bit sfx[4]) {   /* sha3.c:408-436 */
    if (c >= 1600 || ((1600 - c) % 8) != 0) { report("ml_kem shim - sha3_b()", "unsupported capacity"); return NULL; }
    const unsigned rate = (1600 - c) / 8;
    /* sha3.c:414-429: four suffix bits when sfx[2] is set (the XOFs' 1111), two otherwise (hash 01, RawSHAKE 11);
     * the caller's bits are appended verbatim */
    const unsigned nsfx = sfx[2].b == 1 ? 4u : 2u;
    unsigned char sbits[4];
    for (unsigned i = 0; i < nsfx; i++) sbits[i] = (unsigned char)(sfx[i].b & 1u);
    const size_t cap = ((size_t)n + 6 + 8 * rate) / (8 * rate) * rate + rate;
    unsigned char* bits = (unsigned char*)malloc(n ? n : 1);
    unsigned char* padded = (unsigned char*)malloc(cap);
    unsigned char* out = (unsigned char*)malloc((d + 7) / 8 + 1);
    union bit* D = NULL;
    if (bits && padded && out) {
        for (unsigned int i = 0; i < n; i++) bits[i] = (unsigned char)(bstr[i].b & 1u);
        int nblocks = mlkem_sha3_pad_suffix(bits, n, sbits, nsfx, rate, padded, cap);
        int rc = nblocks > 0 ? mlkem_keccak_sponge(rate, 1, padded, (unsigned)nblocks, out, (d + 7) / 8) : nblocks;
        if (!engine_failed("ml_kem shim - sha3_b()", rc)) {
            D = (union bit*)calloc(d ? d : 1, sizeof(union bit));
            if (D) for (unsigned int i = 0; i < d; i++) D[i].b = (out[i / 8] >> (i % 8)) & 1u;
        }
    }
    free(bits); free(padded); free(out);
    return D;
}

union hex* sha3_h(const union hex* hstr, unsigned int m, unsigned int d, unsigned int c, union bit sfx[4]) {   /* sha3.c:443-457 */
    union bit* N = h2b(hstr, m, 8 * m);
    if (!N) return NULL;
    union bit* M = sha3_b(N, 8 * m, d, c, sfx);
    free(N);
    if (!M) return NULL;
    union hex* D = b2h(M, d);
    free(M);
    return D;
}

unsigned char* sha3_s(const char* cstr, unsigned int m, unsigned int d, unsigned int c, union bit sfx[4]) {   /* sha3.c:465-494 */
    union hex* H = (union hex*)calloc(m ? 2 * (size_t)m : 1, sizeof(union hex));
    if (!H) return NULL;
    for (unsigned int i = 0; i < m; i++) {
        H[2 * i].d = ((unsigned char)cstr[i] >> 4) & 15u;
        H[2 * i + 1].d = (unsigned char)cstr[i] & 15u;
    }
    union hex* Z = sha3_h(H, m, d, c, sfx);
    free(H);
    if (!Z) return NULL;
    unsigned char* D = (unsigned char*)malloc(d / 8 ? d / 8 : 1);
    if (D) for (unsigned int i = 0; i < d / 8; i++) D[i] = (unsigned char)((Z[2 * i].d << 4) ^ Z[2 * i + 1].d);
    free(Z);
    return D;
}
