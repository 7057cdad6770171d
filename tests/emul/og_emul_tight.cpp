// TEST-ONLY host emulation of the reconstruction kernel's 8 KB LDS layout (og_recon.hip: OG_RECON_TIGHT): CELT-only 20 ms
// frames through parse (per lane) -> k_celt_recon_fb's code -> post, one-lane "wave".  Frames that kernel would leave to
// the general one come back as -999.  Same entry points as og_emul.cpp so that tools/fuzz_emul_celt.py can drive either.
#define OG_HOST_EMUL 1
#define OG_RECON_TIGHT 1
#include "og_celt_split.hpp"

extern "C" {
int emu_state_size(void) { return (int)sizeof(og::StreamState); }
void emu_stream_init(void *stv, int channels) {
    og::StreamState *st = (og::StreamState *)stv;
    memset(st, 0, sizeof(*st));
    for (int i = 0; i < 2 * og::NBANDS; i++) st->celt.logE1[i] = st->celt.logE2[i] = (int16_t)(-28 * 1024);
    st->channels = channels;
}
int emu_decode_frame(void *stv, const uint8_t *payload, int len, int mode, int bw, int ch, int16_t *pcm) {
    (void)bw;
    if (mode != og::MODE_CELT) return -998;
    og::StreamState *st = (og::StreamState *)stv;
    static og::ParseRec rec;
    og::parse_tables_load();
    og::celt_parse_lane(st, payload, len, ch, &rec, nullptr);
    int ret = og::celt_recon_wave(st, &rec, mode, ch, og::RECON_FAST_ONLY);
    if (ret == og::RECON_NOT_MINE) { // what the general kernel does with such a frame is not this library's business
        if (rec.flags & (og::RF_SKIP | og::RF_BAD_CELT)) return rec.ret;
        return -999;
    }
    for (int c = 0; c < st->channels; c++) og::celt_post(st, &rec, ret, c, pcm, nullptr, ch);
    return ret;
}
}
extern "C" void og_emul_tap(int) {}
