// A host program shaped like the reference player's main.cpp (src/main.cpp:264-282, :308): it DEFINES SD_read,
// calls opus_init_decoder() once and then pulls PCM with op_read_stereo(buf, 2048) until ret <= 0.
// Built against include/opusfile.h and linked with libopusgpu.so by tests/test_container.py (-m gpu).
#include <stdio.h>
#include "opusfile.h"

static FILE *g_in;
static int16_t m_outBuff[2048 * 2];

int SD_read(unsigned char *buff, int nbytes) { // same contract as the reference's SD card reader
    int n = (int)fread(buff, 1, (size_t)nbytes, g_in);
    return n > 0 ? n : -1;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    g_in = fopen(argv[1], "rb");
    FILE *out = fopen(argv[2], "wb");
    if (!g_in || !out) return 2;
    if (!opus_init_decoder()) { printf("init failed\n"); return 1; }
    int calls = 0, ret;
    long total = 0;
    while ((ret = op_read_stereo(m_outBuff, 2048)) > 0) {
        fwrite(m_outBuff, sizeof(int16_t), (size_t)ret * 2, out);
        calls++;
        total += ret;
    }
    printf("calls=%d samples=%ld final=%d\n", calls, total, ret);
    opus_close_decoder();
    fclose(out);
    return 0;
}
