/* oc_celt_priv.h -- CPU ORACLE (test infrastructure): internals shared by the CELT files. */
#ifndef OC_CELT_PRIV_H
#define OC_CELT_PRIV_H
#include "oc_opus.h"
#include "rom_tables.h"

u32 oc_isqrt32(u32 val);
i16 oc_rsqrt_norm(i32 x);
i32 oc_sqrt(i32 x);
i16 oc_cos_norm(i32 x);
i32 oc_rcp(i32 x);
i32 oc_exp2_frac(i32 x);
i32 oc_exp2(i32 x);
u32 oc_pvq_v(int n, int k);

#define BITRES 3

#endif
