// Small C exports of host-side start-up code for Python callers (bench.py / tests build synthetic cohorts in memory
// and need exactly the histogram conditioning that `popdel call` applies when it loads a profile header).
#include <chrono>
#include <cstring>
#include <random>
#include <vector>

#include "call_params.h"
#include "histogram.h"
#include "window_call.h"

extern "C" {

// In: raw counts `values[n]` with table offset / median / stddev / read length. Out: conditioned densities written
// back into `values`, plus min_prob and the two quantile distances.
int pdh_condition_histogram(double * values, unsigned n, int offset, unsigned median, double stddev, unsigned readLength,
                            int smoothing, unsigned pseudoCountFraction, double * minProbOut, unsigned * lowerQOut,
                            unsigned * upperQOut)
{
    if (!values || n < 3 || !minProbOut || !lowerQOut || !upperQOut)
        return 1;
    pdh::InsertHistogram h;
    h.values.assign(values, values + n);
    h.offset = offset;
    h.median = median;
    h.stddev = stddev;
    h.readLength = readLength;
    pdh::conditionHistogram(h, 256, smoothing != 0, pseudoCountFraction);
    std::memcpy(values, h.values.data(), n * sizeof(double));
    *minProbOut = h.minProb;
    *lowerQOut = h.lowerQuantileDist;
    *upperQOut = h.upperQuantileDist;
    return 0;
}

// Micro-benchmark of the two host passes behind the kernels (tools/bench_host_post.py): `windows` synthetic window calls
// of one merge group with `samples` sample columns each are merged (unifyCalls) and the surviving record is formatted
// `records` times. Returns seconds in out[0] (merge) and out[1] (format), bytes of one formatted record in out[2].
int pdh_bench_merge_and_format(unsigned samples, unsigned windows, unsigned records, double * out)
{
    if (!out || samples == 0 || windows < 2)
        return 1;
    std::mt19937 gen(12345);
    auto rng = [&gen]() { return static_cast<uint32_t>(gen()); };
    std::vector<pdh::WindowCall> calls(windows);
    for (unsigned w = 0; w < windows; ++w)
    {
        pdh::WindowCall & c = calls[w];
        c.initialLength = 1000;
        c.iterations = 3;
        c.deletionLength = 1000 + (w % 3);
        c.likelihoodRatio = 500.0 + w;
        c.frequency = 0.2;
        c.windowPosition = 1000000 + 30 * w;
        c.position = 1000010;
        c.endPosition = 1001020;
        c.filter = 0;
        c.pl.resize(samples);
        c.lad.resize(samples);
        c.dad.resize(samples);
        c.firstLast.resize(samples);
        for (unsigned s = 0; s < samples; ++s)
        {
            c.pl[s] = {rng() % 300u, rng() % 50u, rng() % 900u};
            c.pl[s][rng() % 3u] = 0;
            c.lad[s] = {rng() % 30u, rng() % 5u, rng() % 20u};
            c.dad[s] = {rng() % 30u, rng() % 5u, rng() % 3u, rng() % 20u, rng() % 2u};
            c.firstLast[s] = {999000u + rng() % 500u, 1000500u + rng() % 500u};
        }
    }
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    pdh::unifyCalls(calls, 50.0, 0.5, false);
    const auto t1 = clock::now();
    out[0] = std::chrono::duration<double>(t1 - t0).count();
    out[1] = 0;
    out[2] = 0;
    if (calls.empty())
        return 0;
    const pdh::QuantileTable q(0.0001);
    size_t bytes = 0;
    const auto t2 = clock::now();
    for (unsigned r = 0; r < records; ++r)
        bytes = pdh::formatRecord(q, calls[0], "chr20").size();
    out[1] = std::chrono::duration<double>(clock::now() - t2).count();
    out[2] = static_cast<double>(bytes);
    return 0;
}

}  // extern "C"
