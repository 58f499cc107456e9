/* fuzz_driver.c -- sanitizer run of the CPU oracle (test infrastructure).  Built with
 * -fsanitize=address,undefined together with alac_oracle.c and the synthetic encoder; feeds the oracle
 * (a) pure garbage, (b) valid packets with random bit flips / truncation, and checks nothing but
 * "no sanitizer report, status in range".  Shift-count and signed-overflow UB are real traps in a
 * restatement of C# int arithmetic. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "alac_oracle.h"
#include "../alac.net_amd/synth/alac_synth.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 300;
    static int32_t pcm[16384 * 2 + 16];
    static int32_t src[4096 * 2];
    static uint8_t pkt[1 << 17];
    int32_t ob, os;
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        alac_oracle_cfg cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.max_samples_per_frame = (rnd() % 3 == 0) ? 64 : 4096;
        cfg.sample_size = (rnd() & 1) ? 16 : 24;
        if (rnd() % 16 == 0) cfg.sample_size = 20;
        cfg.rice_history_mult = (uint8_t)rnd();
        cfg.rice_initial_history = (uint8_t)rnd();
        cfg.rice_kmodifier = (uint8_t)(rnd() % 17);
        cfg.num_channels = (uint8_t)(1 + (rnd() & 1));
        size_t size;
        if (it & 1) { /* garbage */
            size = 1 + rnd() % 600;
            for (size_t i = 0; i < size; i++) pkt[i] = (uint8_t)rnd();
            pkt[0] &= 0x3F;
        } else { /* valid packet, then mutated */
            alac_synth_pkt d;
            memset(&d, 0, sizeof d);
            d.n = 1 + rnd() % 700; d.max_samples_per_frame = cfg.max_samples_per_frame;
            d.sample_size = cfg.sample_size == 20 ? 16 : cfg.sample_size; d.stereo = cfg.num_channels == 2;
            d.ub = d.sample_size == 24 ? rnd() % 3 : 0;
            d.pred_order[0] = rnd() % 32; d.pred_order[1] = rnd() % 32;
            d.quant[0] = rnd() % 16; d.quant[1] = rnd() % 16; d.ricemod[0] = rnd() % 8; d.ricemod[1] = rnd() % 8;
            d.mix_shift = rnd() % 9; d.mix_weight = rnd() % 2;
            d.rice_history_mult = 40; d.rice_initial_history = 10; d.rice_kmodifier = 14; d.channels_field = -1;
            alac_synth_signal sig = {it, 12.f, 15.f, 500.f, 0.3f, 16, 200, 0.8f};
            alac_synth_make_pcm(&sig, (uint64_t)it, d.sample_size, d.stereo ? 2 : 1, d.n, src);
            size = alac_synth_encode_packet(&d, src, pkt, sizeof pkt);
            if (size == 0) continue;
            int flips = rnd() % 6;
            for (int f = 0; f < flips; f++) pkt[rnd() % size] ^= (uint8_t)(1u << (rnd() & 7));
            if (rnd() % 5 == 0) size = 1 + rnd() % size;
        }
        int st = alac_oracle_decode_frame(&cfg, pkt, size, pcm, 16384 * 2 + 8, &ob, &os);
        if (st < 0 || st > 7) bad++;
    }
    printf("fuzz iterations=%d bad_status=%d\n", iters, bad);
    return bad ? 1 : 0;
}
