#include "oc_opus.h"
struct oc_silk { int x; };
int oc_silk_sizeof(void){return sizeof(struct oc_silk);}
void oc_silk_init(oc_silk*s){(void)s;}
int oc_silk_decode(oc_silk *s, oc_rc *rc, int channels, int internal_hz, int first, i16 *out, i32 *n_out){return -1;}
