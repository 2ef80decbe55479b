/*
 * ml_kem_shim.c — libml_kem.so: the reference's public C API (ml_kem.h) on top of the MI355X batch engine.
 *
 * Each call widens/narrows between the reference's 4-byte `union byte` cells and the packed bytes of the
 * batch C-ABI (include/mlkem_batch.h) and runs a batch of one on the GPU.  There is no CPU fallback: if no
 * HIP device is usable the call prints an error, sets ml_errno = -100 and returns a zeroed result.
 * Error codes, messages and ownership follow ml_kem.c (init :1363, KEM_KeyGen :1233, KEM_Encaps :1257,
 * KEM_Decaps :1310); unlike the reference, error paths return zeroed structs instead of uninitialised ones.
 */
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
    free(ek); free(dk);
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
    free(pdk); free(pc);
    return result;
}
