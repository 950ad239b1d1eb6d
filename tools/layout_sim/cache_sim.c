/* cache_sim.c — a set-associative LRU cache fed by many interleaved access streams (tools/layout_sim/layout_sim.py).
 * One stream = the records one lane of the trace kernel fetches, in order; the streams advance round-robin, one record per
 * turn, which is how ~49,000 resident lanes of an XCD share its 4 MB L2.  Counts lines, not bytes. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t line_accesses, line_hits;      /* 128-byte lines touched / found */
    uint64_t records, record_lines;         /* records fetched, lines they covered (record_lines / records = lines per record) */
    uint64_t warm_accesses, warm_hits;      /* the same after the first `warm` fraction of the rounds */
    uint64_t distinct_lines;                /* lines ever touched */
} SimOut;

int simulate(const uint64_t* starts, int n_streams, const uint64_t* addr, const uint32_t* bytes, int line_bytes, int cache_bytes, int ways,
             double warm, SimOut* out)
{
    const int n_lines = cache_bytes / line_bytes, n_sets = n_lines / ways;
    uint64_t* tag = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n_lines);
    uint64_t* age = (uint64_t*)calloc((size_t)n_lines, sizeof(uint64_t));
    uint64_t* cur = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n_streams);
    if (!tag || !age || !cur) return 1;
    memset(tag, 0xff, sizeof(uint64_t) * (size_t)n_lines);
    memset(out, 0, sizeof(*out));
    uint64_t longest = 0, clock = 0;
    for (int s = 0; s < n_streams; s++) {
        cur[s] = starts[s];
        if (starts[s + 1] - starts[s] > longest) longest = starts[s + 1] - starts[s];
    }
    /* distinct lines: a bitmap over the address range seen */
    uint64_t max_line = 0;
    for (uint64_t i = 0; i < starts[n_streams]; i++) {
        uint64_t l = (addr[i] + bytes[i]) / (uint64_t)line_bytes;
        if (l > max_line) max_line = l;
    }
    uint8_t* seen = (uint8_t*)calloc((size_t)(max_line / 8 + 2), 1);
    if (!seen) return 1;
    const uint64_t warm_round = (uint64_t)(warm * (double)longest);
    for (uint64_t round = 0; round < longest; round++) {
        for (int s = 0; s < n_streams; s++) {
            const uint64_t i = cur[s];
            if (i >= starts[s + 1]) continue;
            cur[s] = i + 1;
            const uint64_t l0 = addr[i] / (uint64_t)line_bytes, l1 = (addr[i] + bytes[i] - 1) / (uint64_t)line_bytes;
            out->records++;
            for (uint64_t l = l0; l <= l1; l++) {
                out->record_lines++;
                out->line_accesses++;
                if (round >= warm_round) out->warm_accesses++;
                if (!(seen[l >> 3] & (1u << (l & 7)))) { seen[l >> 3] |= (uint8_t)(1u << (l & 7)); out->distinct_lines++; }
                const size_t base = (size_t)(l % (uint64_t)n_sets) * (size_t)ways;
                int hit = -1, victim = 0;
                for (int w = 0; w < ways; w++) {
                    if (tag[base + w] == l) { hit = w; break; }
                    if (age[base + w] < age[base + victim]) victim = w;
                }
                clock++;
                if (hit >= 0) {
                    age[base + hit] = clock;
                    out->line_hits++;
                    if (round >= warm_round) out->warm_hits++;
                } else {
                    tag[base + victim] = l;
                    age[base + victim] = clock;
                }
            }
        }
    }
    free(tag); free(age); free(cur); free(seen);
    return 0;
}
