/*
 * popdel_hip.h -- C ABI of the MI355X joint-calling backend (libpopdel_hip.so).
 *
 * This is the drop-in boundary for PopDel's `popdel call` hot path. The reference has no
 * FFI; its seam is the C++ call
 *     bool genotype_deletion_window(String<Call>&, const ChromosomeProfile&, const TRGs&,
 *                                   PopDelCallParameters&)
 * (reference popdel_call/genotype_deletion_popdel_call.h:536-539) invoked once per 30-bp
 * window from processSegment() (reference workflow_popdel.h:42-47, :73-80). A per-window
 * call cannot feed a GPU, so the boundary is moved up one loop level: one call per
 * *segment* (all windows walked by one processSegment() invocation). Everything the window
 * loop reads from ChromosomeProfile is handed over as plain arrays, everything it appends
 * to String<Call> comes back as plain arrays.
 *
 * Conventions: all functions return 0 on success, non-zero on error (pd_last_error() gives
 * text); no exceptions cross the boundary; a context is bound to one GPU and calls on one
 * context must be serialised by the caller. Tickets: at most two PER LANE may be outstanding per context
 * (a context has three lanes unless POPDEL_HIP_LANES says otherwise; a plain context -- one lane -- and a
 * cohort sharded over several ranks take two: one running, one waiting), collected in the order issued;
 * see pd_segment_run(). Pointers
 * passed in are caller-owned host memory unless stated otherwise; pd_calls is callee-owned
 * until pd_calls_free().
 *
 * Limits (violations are reported as errors, never truncated silently):
 *   - at most 65,535 samples per context (pd_ctx_create fails beyond; larger cohorts are sharded
 *     by samples over several contexts, pd_ctx_set_shard);
 *   - at most 64 EM iterations (pd_config::iterations);
 *   - at most 32 candidate deletion lengths per window (pd_segment_collect fails beyond; the
 *     reference's clustering yields one per ~50 bp of Q3 spread, a handful at most in practice);
 *   - segments of at most 2^31 windows and 2^32 - 1 pair chunks per read group.
 */
#ifndef POPDEL_HIP_H_
#define POPDEL_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_ABI_VERSION 14

/* One read group after start-up conditioning.
 * Replaces: `Histogram` (reference insert_histogram_popdel.h:19-43) after processHistogram()
 * (:974-986), plus the per-read-group entries of PopDelCallParameters::maxLoad and
 * ::minInitDelLengths (reference parameter_parsing_popdel_call.h:153,177). */
typedef struct pd_readgroup
{
    uint32_t      median;               /* Histogram::median                       */
    uint32_t      read_length;          /* Histogram::readLength                   */
    double        stddev;               /* Histogram::stddev                       */
    double        min_prob;             /* Histogram::min_prob                     */
    int32_t       hist_offset;          /* Histogram::offset                       */
    uint32_t      hist_len;             /* length(Histogram::values)               */
    const double *hist_values;          /* conditioned densities, hist_len entries */
    uint32_t      lower_quantile_dist;  /* Histogram::lowerQuantileDist            */
    uint32_t      upper_quantile_dist;  /* Histogram::upperQuantileDist            */
    uint32_t      max_load;             /* PopDelCallParameters::maxLoad[rg]       */
    uint32_t      min_init_del_len;     /* PopDelCallParameters::minInitDelLengths[rg] */
} pd_readgroup;

/* Scalar calling parameters read inside the hot path.
 * Replaces: the members of PopDelCallParameters used by genotype_deletion_window()
 * (reference genotype_deletion_popdel_call.h:548,583,598,690,692,723). */
typedef struct pd_config
{
    uint32_t iterations;              /* params.iterations (default 15)            */
    uint32_t min_len;                 /* params.minLen                             */
    double   min_likelihood_ratio;    /* params.minimumLikelihoodRatio             */
    double   min_sample_fraction;     /* params.minSampleFraction                  */
    int32_t  somatic;                 /* params.somatic (-C)                       */
    int32_t  window_wise_output;      /* params.windowWiseOutput (-n)              */
} pd_config;

/* One segment = the windows walked by one processSegment() call.
 * Replaces: the ChromosomeProfile contents read by the window loop (activeReads, startProfiles,
 * currentPos; reference profile_structure_popdel_call.h:1035-1060). A read pair that is a member
 * of ChromosomeProfile::activeReads[rg] for the windows with index on <= w < off appears as one
 * "pair instance" {dev, start, on, off}. The host side derives on/off by running the
 * reference's window state machine, so the loader's segment-border behaviour stays host-side.
 * Instances of one read group are stored contiguously, ordered by non-decreasing `on`. */
typedef struct pd_segment
{
    uint32_t        first_window_pos;  /* ChromosomeProfile::currentPos of window 0          */
    uint32_t        n_windows;         /* number of genotype_deletion_window() calls         */
    uint32_t        window_shift;      /* params.windowShift (30)                            */
    uint32_t        n_readgroups;      /* must equal the context's read-group count          */
    const uint64_t *rg_offsets;        /* n_readgroups + 1 offsets into the pair arrays      */
    const int32_t  *pair_dev;          /* ChromosomeProfileStartEntry::insertSizeDeviation   */
    const uint32_t *pair_start;        /* ChromosomeProfileStartEntry::startPos              */
    const uint32_t *pair_on;           /* first window index in which the pair is active     */
    const uint32_t *pair_off;          /* first window index in which it is no longer active */
} pd_segment;

/* Calls of one segment, in the order the reference appends them to String<Call>
 * (window ascending, initial candidate length ascending within a window).
 * Replaces: `Call` (reference utils_popdel.h:57-124) as filled at
 * genotype_deletion_popdel_call.h:692-724. Per-sample arrays are [n_calls][n_samples][k]. */
typedef struct pd_calls
{
    uint32_t  n_calls;
    uint32_t  n_samples;
    uint32_t *initial_length;     /* Call::initialLength   */
    uint32_t *iterations;         /* Call::iterations      */
    uint32_t *deletion_length;    /* Call::deletionLength  */
    double   *likelihood_ratio;   /* Call::likelihoodRatio */
    double   *frequency;          /* Call::frequency       */
    uint32_t *window_position;    /* Call::windowPosition  */
    uint32_t *position;           /* Call::position        */
    uint32_t *end_position;       /* Call::endPosition     */
    uint8_t  *filter;             /* Call::filter          */
    uint32_t *pl;                 /* Call::gtLikelihoods, 3 per sample (ref, het, hom)              */
    uint32_t *lad;                /* Call::lads, 3 per sample                                       */
    uint32_t *dad;                /* Call::dads, 5 per sample (ref, both, between, alt, right)      */
    uint32_t *first_last;         /* Call::firstLast, 2 per sample                                  */
    /* Parity instrumentation (ABI 12; no reference counterpart): per call, how many reference-shift quotients of ITS
     * EM (all samples, read groups and iterations; reference genotype_deletion_popdel_call.h:447-450) sat within 1e-9
     * of an integer -- the truncations whose outcome hangs off the last bit of x87 weights (pd_timing below, DESIGN.md
     * "Numerics"). A call that differs from the reference's anywhere has a non-zero count; the results are untouched. */
    uint32_t *near_integer_shifts;
} pd_calls;

/* Per-window stage-1 results of the last collected segment (diagnostics / parity tests).
 * q3[w * n_samples + s] is the value upperHalfMedian() returned for sample s in window w,
 * flags bit0 = lowCoverageSamples[s], bit1 = sample contributed a Q3 value.
 * Replaces nothing in the reference's interface; mirrors the locals of
 * initialize_deletion_lengths() (reference genotype_deletion_popdel_call.h:33-57). */
typedef struct pd_stage1_view
{
    uint32_t       n_windows;
    uint32_t       n_samples;
    const int32_t *q3;
    const uint8_t *flags;
} pd_stage1_view;

typedef struct pd_ctx pd_ctx;

/* pd_stage1_last() needs the Q3 of EVERY cell; since ABI 12 the HIP backend screens the windows first (counts, per cell, the
 * active pairs whose deviation exceeds the smallest threshold -- which decides whether the window can carry a candidate at
 * all) and takes exact Q3 values only in the windows that may, so the values of the other windows do not exist unless
 * asked for: pd_ctx_stage1_view(ctx, 1), before a segment is run, makes the context ALSO run the full scan (exact Q3 of
 * every cell, the round-1..4 kernel) for that purpose. pd_stage1_last() then returns the full scan's values after checking
 * them against the product path cell by cell (flags everywhere; every window with a Q3 above the threshold flagged by the
 * screen; the exact pass's Q3 equal in every flagged window) and fails if they disagree. Diagnostics and parity tests only;
 * costs a second scan per segment. The oracle computes every cell anyway (the call is accepted and changes nothing). */
int pd_ctx_stage1_view(pd_ctx *ctx, int on);

/* Create a calling context on GPU `device`.
 * sample_rg_offsets has n_samples + 1 entries; sample s owns read groups
 * [sample_rg_offsets[s], sample_rg_offsets[s+1]) (reference TRGs, utils_popdel.h:21-22; read groups
 * of one profile file are numbered consecutively, insert_histogram_popdel.h:1029-1037).
 * Replaces: PopDelCallParameters + String<Histogram> after loadAndCalculateParameters()
 * (reference parameter_calculation_popdel_call.h:160-204). */
int pd_ctx_create(const pd_config *cfg, const pd_readgroup *rgs, uint32_t n_readgroups,
                  const uint32_t *sample_rg_offsets, uint32_t n_samples, int device, pd_ctx **out);

void pd_ctx_destroy(pd_ctx *ctx);

/* Stage a segment into HBM and enqueue the whole window loop for it. Returns a ticket.
 * Replaces: the `while (!profilesAtEnd) genotype_deletion_window(...)` loop of processSegment()
 * (reference workflow_popdel.h:42-47). */
int pd_segment_submit(pd_ctx *ctx, const pd_segment *seg, uint64_t *ticket);

/* The two halves of pd_segment_submit(), for callers that keep segments resident in HBM or stream them double
 * buffered: pd_segment_stage() copies a segment into HBM (returns a handle), pd_segment_run() enqueues the window
 * loop for a staged segment (may be repeated), pd_segment_unstage() releases it.
 * pd_segment_submit(seg) == stage + run + unstage-after-collect. No reference counterpart (the reference computes
 * in place on the host).
 * Tickets: at most two per lane may be outstanding per context, collected in the order issued. Issuing the next ticket
 * BEFORE collecting the open one lets the library start the next segment's scan right behind the open segment's last
 * kernel, underneath the device-to-host copy of its results (software pipelining across segments); with lanes
 * (POPDEL_HIP_LANES, default 3: child contexts that take the tickets in turn, each driven by a host thread of the
 * library's) several segments' stage 2 run side by side and fill each other's gaps -- a host that wants the context's
 * full rate keeps 2 x lanes tickets in flight. With columns kept on the device (pd_ctx_keep_results_on_device) one
 * ticket per lane. */
int pd_segment_stage(pd_ctx *ctx, const pd_segment *seg, uint64_t *handle);
int pd_segment_run(pd_ctx *ctx, uint64_t handle, uint64_t *ticket);
int pd_segment_unstage(pd_ctx *ctx, uint64_t handle);

/* Device time of the kernels of the last collected ticket, measured with HIP events on the launch stream.
 * No reference counterpart (instrumentation). */
typedef struct pd_timing
{
    double   stage1_ms;          /* Q3 scan kernel                                            */
    double   stage2_ms;          /* clustering + EM + final pass kernels                      */
    uint64_t n_cells;            /* windows x samples handled by the scan                     */
    uint64_t n_active_pairs;     /* sum over cells of active pairs (algorithmic gather size)  */
    uint64_t n_pair_chunks;      /* pair chunks resident in HBM for the segment               */
    uint64_t n_candidate_cells;  /* (window, candidate length) cells entering the EM          */
    uint32_t max_life;           /* chunk length K in windows                                 */
    uint32_t n_all_gathers;      /* exchanges of the ticket (cohort sharded by samples; else 0) */
    uint64_t all_gather_bytes;   /* bytes gathered per context by them                        */
    uint64_t n_repeated_batches; /* stage-2 batches of this context (cumulative) that were launched before the host had
                                    seen their support-pool sizes and had to be run again the exact way */
    double   exchange_ms;        /* time the ticket's exchanges took: HIP events around each on the context's stream (the
                                    in-process group: host clock around its barriers and peer copies); part of stage2_ms */
    double   lik_first_ms;       /* the data-likelihood kernels' launch ahead of the EM loop (last stage-2 batch): all cells of */
    uint64_t lik_first_units;    /* the batch x all samples of the context = this many (cell, sample) units                   */
    /* Parity instrumentation (ABI 11). A reference shift is int(sum(p * dev) / sum(p)) (reference
     * genotype_deletion_popdel_call.h:447-450); where the quotient sits within 1e-9 of an integer the truncation is decided by
     * the last bit of weights that the reference forms in x87 extended precision and this backend in FP64 -- the one place
     * where the two may take different decisions (Numerics). Counted per ticket over all cells, EM iterations and
     * read groups of the COHORT (all contexts of a sharded cohort report the same number); the results are not changed. */
    uint64_t n_near_integer_shifts;
} pd_timing;
int pd_last_timing(pd_ctx *ctx, pd_timing *out);

/* What the last pd_segment_stage() / pd_segment_submit() of the context cost on the host side and on the link (ABI 12;
 * instrumentation, no reference counterpart): the pair instances cross PCIe once in a compact form that the library's host
 * threads pack from the caller's four 32-bit arrays (16 B per instance read from the caller's memory, `flavour` bytes per
 * instance written to page-locked staging and uploaded slab by slab while the next slab is packed).
 *   pack_seconds      wall time of the packing passes (sum over the slabs)
 *   uploaded_seconds  from the start of the call until the last uploaded byte had reached the device
 *   bytes_uploaded    n_instances x flavour (+ the small tables)
 * pd_host_copy_gbs(bytes): a probe of what the host's memory system gives those threads right now -- `bytes` copied from
 * pageable to page-locked memory by the same threads in the same way, GB/s (read + write counted once). The streaming rate of
 * a context is bound by the packing pass, i.e. by this figure, not by PCIe (streaming). */
typedef struct pd_stage_timing
{
    uint64_t n_instances;
    uint64_t bytes_uploaded;
    double   pack_seconds;
    double   uploaded_seconds;
    uint32_t flavour;            /* bytes per instance on the link: 5 (compact), 8 (narrow), 16 (wide); 0: built on the host */
    uint32_t host_threads;
    double   counted_seconds;    /* from the start of the call until the device's counting pass had reported the layout's size
                                    (the build's second and last wait for the device; the scatter that follows is not waited for) */
    double   total_seconds;      /* the whole call */
} pd_stage_timing;
int pd_last_stage_timing(pd_ctx *ctx, pd_stage_timing *out);
double pd_host_copy_gbs(uint64_t bytes);

/* ---- cohort sharded by samples over several contexts (one per GPU) -------------------------------------------------
 * The reference has no counterpart (single process); this is how `genotype_deletion_window()` is split over GPUs:
 * context r of `world` owns the samples [first_sample, first_sample + n_samples) of the cohort, where the partition is
 * the one pd_shard_samples() returns (whole 256-sample tiles per context). Every context is created with ITS OWN
 * samples/read groups only (pd_ctx_create), receives ITS OWN pairs in pd_segment, and returns the call list of the
 * whole cohort with the per-sample columns of its own samples. What crosses samples -- the sorted Q3 values per
 * flagged window (reference :58-87), the integer counts of :93-133, per EM iteration the sums of :388-462 / :467-485 and
 * read group 0's triple (quirk: :411), the likelihood-ratio sums of :490-508 and the support percentiles of :514-529 --
 * is all-gathered between kernels through `all_gather` and reduced in a fixed order, so the results are bit-identical to
 * the unsharded run. All contexts of a cohort must call pd_segment_run / pd_segment_collect for the same segments in
 * the same order (the calls are collective).
 *
 * reserve():    returns a device buffer of at least `bytes` (contents undefined; the previous buffer is dead). The host
 *               owns it (e.g. a torch tensor), so that it can address the regions in its collective library.
 * all_gather(): the region [offset, offset + world * bytes_per_rank) of the current buffer holds `world` chunks; chunk
 *               `rank` is filled. On return (or stream-ordered on `hip_stream`) all chunks must be filled on every
 *               context. Return non-zero on failure. */
typedef struct pd_shard
{
    uint32_t rank, world;
    uint32_t first_sample;                  /* cohort-wide index of this context's first sample                      */
    uint32_t n_samples_global;
    uint32_t n_readgroups_global;
    const uint32_t *min_init_del_len_global; /* [n_readgroups_global], cohort order (rank-indexed lookup, quirk Q1)    */
    void *(*reserve)(void *user, uint64_t bytes);
    int (*all_gather)(void *user, uint64_t offset, uint64_t bytes_per_rank, void *hip_stream);
    void *user;
} pd_shard;

/* A context with lanes (POPDEL_HIP_LANES, pd_ctx_create) runs its tickets on several child contexts at once; their exchanges
 * must not share a channel (they would interleave in an order no other rank reproduces). pd_ctx_set_shard() therefore runs a
 * cohort over several ranks without lanes. pd_ctx_set_shard_lanes() keeps them: `shard` as above for the context itself (the
 * exchanges of pd_segment_stage), and one pd_shard per lane -- same rank and cohort, its OWN reserve / all_gather / user, i.e.
 * its own buffer and its own group of the caller's collective library; lane k of every rank then exchanges with lane k of the
 * others, from a thread of the library (the callbacks must be callable from any thread). Tickets go to min(lanes of the
 * context, n_lane_shards) lanes, ticket t to lane t mod that number, on every rank alike. Results are bit-identical to the
 * unsharded run (tests/test_gpu_sample_shard.py: two processes over gloo with two lanes). */
int pd_ctx_set_shard_lanes(pd_ctx *ctx, const pd_shard *shard, const pd_shard *lane_shards, uint32_t n_lane_shards);

/* The same with the exchange done by the library itself: RCCL all-gathers issued on the context's own HIP stream, in place
 * in an exchange buffer the library owns (no host callbacks in the data path; `reserve`, `all_gather` and `user` of
 * `shard` are ignored). One process calls pd_rccl_unique_id() and hands the 128 bytes to all others over any channel;
 * then EVERY context of the cohort calls pd_ctx_set_shard_rccl() with them (the call is collective: it creates the
 * communicator). RCCL is loaded at that moment (librccl.so.1; POPDEL_RCCL_LIBRARY names another file), a process that
 * never shards does not need it. */
#define PD_RCCL_ID_BYTES 128
int pd_rccl_unique_id(uint8_t *id /* PD_RCCL_ID_BYTES */);
int pd_ctx_set_shard_rccl(pd_ctx *ctx, const pd_shard *shard, const uint8_t *id /* PD_RCCL_ID_BYTES */);

/* Contexts of ONE process (one host thread each; several GPUs, or several contexts on one GPU): the library exchanges
 * between them itself -- peer copies between the contexts' exchange buffers, bracketed by barriers of the host threads.
 * Create the group, then pd_ctx_set_shard_local() on every context (`reserve`, `all_gather`, `user` of `shard` are
 * ignored); the contexts' collective calls (pd_segment_run / _collect / _submit) must be issued from `world` threads
 * concurrently. An error in one context fails the pending and later collective calls of all. Destroy the group after
 * its contexts. This is how `popdel call` drives several GPUs (POPDEL_DEVICES). */
typedef struct pd_local_group pd_local_group;
int pd_local_group_create(uint32_t world, pd_local_group **out);
void pd_local_group_destroy(pd_local_group *group);
int pd_ctx_set_shard_local(pd_ctx *ctx, const pd_shard *shard, pd_local_group *group);

/* Partition rule: samples per context (a multiple of 256; the last contexts may own fewer or none). */
uint32_t pd_shard_samples(uint32_t n_samples_global, uint32_t world);

/* Must be called before the first segment is run. */
int pd_ctx_set_shard(pd_ctx *ctx, const pd_shard *shard);

/* ---- segments handed over as the loader's tables hold them: the window loop runs in closed form on the device ----------
 * pd_segment asks the host for every pair's window interval, i.e. for a walk over all windows of the segment with its
 * activations and deactivations (the reference's ChromosomeProfile::initializeActiveReads / nextWindow / nextStartWindow /
 * nextEndWindow, profile_structure_popdel_call.h:1120-1213, as processSegment() drives them, workflow_popdel.h:58-86).
 * What that walk does to a pair is a function of a few numbers per segment and, per pair, of its start, its last window and
 * the end slot its end record was filed under ("The window loop in closed form"). With pd_walk the host hands
 * over, per read group, the start entries of the slot the loader has just switched to reading (start position and
 * deviation, in the order they were filed -- ChromosomeProfileStartEntry, :15-25) and plans the segment's shape; the
 * library derives the pair instances itself and carries from segment to segment what the reference's active sets and
 * read cursors carry: the pairs still active, and the start entries no window has reached yet. Segments must be submitted
 * in the order the loader produces them, every read group in every one of them (its `switches` says what the loader did
 * to its slots since the previous segment).
 *   first_window_pos  position of window 0 (ChromosomeProfile::currentPos after initializeActiveReads)
 *   step_pos/has_step position of the loader's extra nextWindow() step ahead of window 0 (workflow_popdel.h:64-65), if taken
 *   switches[rg]      0: the read group's slots were not switched since the previous segment; 1: once (performSwitches /
 *                     performPartialSwitches, :1878-1910); 2: twice or more, or the tables were reset (resetTo, :1717)
 *   end_border[rg]    right border of the END slot the new entries' end records were filed under: a pair whose last window
 *                     lies at or beyond it was filed one slot ahead (CyclicEndEntryTable::add, :760-790)
 *   id_offset[rg]     the reference's id of the read group's first new entry (only reported back, pd_walk_active)
 *   equal_run[rg]     first segment after a reset: the run of entries on first_window_pos is active in window 0 (:1143-1158)
 */
typedef struct pd_walk_entry    /* ChromosomeProfileStartEntry (profile_structure_popdel_call.h:15-25) */
{
    uint32_t start;             /* startPos */
    int32_t  dev;               /* insertSizeDeviation */
} pd_walk_entry;

typedef struct pd_inflater pd_inflater;     /* (a persistent device gunzip + parser: pd_inflater_create() below) */

/* Where a read group's new entries ALSO lie on the device (ABI 14): records [first_record, first_record + rg_counts[rg]) of the
 * records array the inflater's last pd_inflater_run_parsed() / _summarized() left in HBM -- the loader parsed them there, the host
 * filed them into its table in the same order. If given (inflater != NULL) and the inflater works on the context's GPU, the
 * library takes the entries from there instead of fetching the host's table over the link; rg_entries[rg] must still point to the
 * table (it is what every other case reads) unless the host kept no table for them. The inflater must not run again before
 * pd_walk_submit() returns; run_id lets the library notice if it has. */
typedef struct pd_walk_source
{
    pd_inflater *inflater;
    uint64_t     first_record;
    uint64_t     run_id;        /* pd_inflater_run_id() when the chunk was parsed: the records are taken only if it still is the inflater's last */
} pd_walk_source;

typedef struct pd_walk
{
    uint32_t        first_window_pos;
    uint32_t        step_pos;
    uint32_t        has_step;
    uint32_t        n_windows;         /* > 0 */
    uint32_t        window_shift;
    uint32_t        n_readgroups;      /* must equal the context's read-group count */
    const uint32_t *rg_counts;         /* per read group: entries of its new start slot (ignored where switches == 0) */
    const pd_walk_entry *const *rg_entries;   /* per read group: the entries, in the order they were filed (the host's table itself) */
    const uint32_t *end_border;
    const uint32_t *id_offset;
    const uint8_t  *switches;
    const uint8_t  *equal_run;
    const pd_walk_source *rg_sources;  /* NULL, or per read group where its entries lie on the device as well (see pd_walk_source) */
} pd_walk;

/* Stage the segment (entries -> HBM -> pair instances -> layout) and enqueue the window loop's genotyping for it; the
 * ticket is collected with pd_segment_collect() like that of pd_segment_submit(). */
int pd_walk_submit(pd_ctx *ctx, const pd_walk *walk, uint64_t *ticket);

/* Host address ranges that hold the tables pd_walk::rg_entries point into and stay mapped for the rest of the process (the
 * host's table pool hands them out region by region). The library may page-lock a range the first time a walk's entries lie
 * inside it and from then on lets the DEVICE fetch such entries from the tables themselves (one gather kernel over the link)
 * instead of copying them into a page-locked array on the host threads first -- 8 bytes per pair and segment of host CPU
 * time, a twelfth of a round's in `popdel call`. Ranges may be offered again; entries outside every offered range take the
 * staging copy as before. Must not be called for memory that will be unmapped or given back (MADV_DONTNEED) while a context
 * lives. Returns 0. (ABI 13) */
int pd_host_regions_offer(const void *const *begin, const uint64_t *bytes, uint32_t n);
/* Diagnosis: entries of all pd_walk_submit() calls of the context so far that the device fetched from the host's tables
 * (*gathered) and that went through the staging copy (*staged); bytes of offered ranges page-locked so far (*locked_bytes).
 * Any pointer may be NULL. */
int pd_walk_entry_counts(pd_ctx *ctx, uint64_t *gathered, uint64_t *staged, uint64_t *locked_bytes);

/* ... and how many entries were taken from records already on the device (pd_walk_source; ABI 14). */
int pd_walk_source_counts(pd_ctx *ctx, uint64_t *from_device);

/* 1 if the backend behind the ABI offers pd_walk_submit() (the oracle does not: it takes pd_segment only), else 0. A
 * property of the library: `ctx` may be NULL (a caller that plans its first segment while the context is still being
 * created asks without one). */
int pd_walk_supported(const pd_ctx *ctx);

/* The pairs still active after the most recently submitted walk (waits for its kernels): their number, and -- if the
 * arrays are given -- read group, start, deviation and id of each, read group by read group in id order (room for
 * `capacity` pairs; the return code is 2 if there are more). A host that leaves the closed form (ROI change with pairs
 * still active: the reference keeps them, profile_structure_popdel_call.h:1735-1760) rebuilds its active sets from them. */
int pd_walk_active(pd_ctx *ctx, uint64_t *n_active, uint32_t *rg, uint32_t *start, int32_t *dev, uint32_t *id, uint64_t capacity);
/* ... with, per pair, its last window and -- bit 31 -- whether its end record was filed one end slot AHEAD (ABI 14; `last` may be
 * NULL). A host that takes the window loop back after the loader has switched slots once more (a round was read, its segment never
 * submitted) must do to these pairs what that switch does to the reference's active set: a switch drains the end read slot
 * (profile_structure_popdel_call.h:1878-1910 -> every pair whose end record sits there leaves), so after ONE switch only the pairs
 * filed ahead remain, after two or more none. */
int pd_walk_active_ex(pd_ctx *ctx, uint64_t *n_active, uint32_t *rg, uint32_t *start, int32_t *dev, uint32_t *id, uint32_t *last,
                      uint64_t capacity);

/* Wait for a ticket and hand back its calls (callee-owned; release with pd_calls_free).
 * Replaces: reading `String<Call> calls` after the window loop (reference workflow_popdel.h:48). */
int pd_segment_collect(pd_ctx *ctx, uint64_t ticket, pd_calls **out);

void pd_calls_free(pd_calls *calls);

/* ---- per-sample part of call merging on the device -------------------------------------------------------------------
 * The reference merges the window calls of a segment into variants (unifyCalls, utils_popdel.h:567-654): a scalar pass
 * over call-level fields decides which calls form a variant, then setGenotypes() (:446-501) averages the members'
 * phred-scaled likelihoods and takes per-component upper medians of LAD / DAD per sample, and setFreqFromGTs() (:339-359)
 * counts alleles. With pd_ctx_keep_results_on_device(ctx, 1), pd_segment_collect() returns the call-level fields only
 * (pl, lad, dad, first_last are NULL) and keeps the per-sample arrays in HBM until the next pd_segment_collect() on the
 * context; the host runs the scalar pass and hands the groups to pd_calls_merge(): group g consists of the calls
 * member_index[member_offsets[g] .. member_offsets[g + 1]) (indices into `calls`), head_index[g] is the call whose
 * first_last columns the variant keeps. Returned (callee-owned, pd_merged_free): per group and sample the merged PL /
 * LAD / DAD / first_last, per group the allele count over this context's samples. Return code 2: a LAD / DAD value
 * beyond 127 (the kernel's median bins) -- nothing is returned; fetch the columns and merge on the host.
 * pd_calls_fetch_columns() fills pl / lad / dad / first_last of such a pd_calls after the fact (same lifetime rule).
 * A segment whose stage 2 ran in several batches is returned with its columns regardless of the mode. */
typedef struct pd_merged
{
    uint32_t  n_groups;
    uint32_t  n_samples;
    uint32_t *pl;           /* [n_groups][n_samples][3] */
    uint32_t *lad;          /* [n_groups][n_samples][3] */
    uint32_t *dad;          /* [n_groups][n_samples][5] */
    uint32_t *first_last;   /* [n_groups][n_samples][2] */
    uint32_t *alleles;      /* [n_groups] */
} pd_merged;
int pd_ctx_keep_results_on_device(pd_ctx *ctx, int on);
int pd_calls_merge(pd_ctx *ctx, const pd_calls *calls, uint32_t n_groups, const uint32_t *member_offsets,
                   const uint32_t *member_index, const uint32_t *head_index, pd_merged **out);
int pd_calls_fetch_columns(pd_ctx *ctx, pd_calls *calls);
void pd_merged_free(pd_merged *merged);

/* Stage-1 values of the most recently collected segment; valid until the next submit/run issued AFTER that collect. */
int pd_stage1_last(pd_ctx *ctx, pd_stage1_view *out);

/* The device-resident lookup tables of read group `rg` (diagnostics / parity tests): the conditioned densities as uploaded
 * and their natural and decadic logarithms as the kernels read them (computed once on the device at pd_ctx_create).
 * Each output array takes hist_len entries; returns the number of entries through *n.
 * Replaces nothing in the reference's interface; the reference evaluates logl()/log10l() of Histogram::values per read
 * pair (genotype_deletion_popdel_call.h:205-215, 290-300). */
int pd_log_table_view(pd_ctx *ctx, uint32_t rg, double *value, double *log_value, double *log10_value, uint32_t *n);

/* ---- gunzip of profile blocks on the device (ABI 11) ---------------------------------------------------------------------
 * A profile holds one gzip member per 10,000-bp index region (the writer ends a member per region, reference
 * workflow_popdel.h:201-241; the reader jumps to one through the index, insert_histogram_popdel.h:531-562, and inflates it
 * with zlib, seqan/stream/iostream_zip_impl.h:336-349). The members of a segment are independent: pd_inflate_members()
 * inflates `n_members` of them on GPU `device`, one lane each. data[member_offsets[m] .. member_offsets[m + 1]) is member m
 * (header, deflate stream, trailer); its output goes to out[out_offsets[m] .. out_offsets[m + 1]) -- the caller sizes that from
 * the ISIZE word that ends a member. status[m]: 0, or why the member was refused (1 not gzip/deflate, 2 truncated, 3 bad
 * block header, 4 bad code set, 5 distance before the start, 6 more output than room, 7 CRC-32 mismatch, 8 ISIZE mismatch --
 * the cases zlib's inflate refuses; which of them a damaged member gets is not part of the contract beyond zero / non-zero and 6);
 * out_lengths[m]: bytes produced. All pointers are host memory. Two kernels per launch since round 6 (no ABI change): the
 * decoder, sixteen members per wave, and the CRC-32 of what it produced against the trailers. The kernels are launched
 * `repeat` times (>= 1) and `timing` reports their mean duration next to the two copies. Returns 0, or non-zero with
 * pd_inflate_last_error() saying why (the message of the calling thread's last failed call). This one-shot form serves the
 * bench and the tests; `popdel call` inflates through the persistent pd_inflater objects below (call_driver.cpp,
 * GunzipPrefetch: a chunk of samples' members per call, three inflaters in turn). */
typedef struct pd_inflate_timing
{
    double   h2d_ms;        /* compressed bytes, page-locked host memory -> HBM */
    double   kernel_ms;     /* one launch */
    double   d2h_ms;        /* inflated bytes back (a loader that parses on the device would not pay this) */
    uint64_t in_bytes, out_bytes;
    uint32_t workgroups;    /* of one wave each */
} pd_inflate_timing;
int pd_inflate_members(int device, const uint8_t *data, const uint64_t *member_offsets, uint32_t n_members, uint8_t *out,
                       const uint64_t *out_offsets, uint32_t *status, uint64_t *out_lengths, uint32_t repeat,
                       pd_inflate_timing *timing);
const char *pd_inflate_last_error(void);

/* The same as an object that keeps its stream, device buffers and page-locked staging from call to call (how `popdel call`
 * inflates the profile blocks of a round: a chunk of samples per call, several inflaters in turn). The caller gathers the
 * members into the page-locked buffer pd_inflater_input() returns (room for `bytes`; valid until the next call of it),
 * pd_inflater_run() copies them up, inflates and copies back: *out (page-locked, callee-owned, valid until the next run on
 * this inflater) holds member m's bytes at out_offsets[m], *status / *out_lengths as in pd_inflate_members(). One inflater
 * is used from one thread at a time. pd_inflater_create() returns NULL if the backend has no device (the oracle). */
pd_inflater *pd_inflater_create(int device);
const char *pd_inflater_last_error(pd_inflater *inflater);     /* why this inflater's last run failed (ABI 12) */
void pd_inflater_destroy(pd_inflater *inflater);
uint8_t *pd_inflater_input(pd_inflater *inflater, uint64_t bytes);
int pd_inflater_run(pd_inflater *inflater, const uint64_t *member_offsets, uint32_t n_members, const uint64_t *out_offsets,
                    const uint8_t **out, const uint32_t **status, const uint64_t **out_lengths, double *kernel_ms);

/* ... and with the blocks PARSED on the device as well: what the loader reads out of a block's bytes (reference: readWindow(),
 * popdel_profile/window_podel.h:211-251, per 256-bp window, then convertWindow(), :314-418 -- the 256 -> 30 split). For every block
 * m with parse_member[m] != 0 (a file with ONE read group) whose bytes inflate and parse without anything the host's reader would
 * report, the device leaves
 *   profile_windows[profile_window_offsets[m] ..[m + 1])   the block's 256-bp windows {chrom, beginPos, splitFailed = 0, first 30-bp
 *                                                          window, number of 30-bp windows} (20 bytes each)
 *   call_windows[call_window_offsets[m] ..)                its 30-bp windows with content {chrom, beginPos, first record} (12 bytes)
 *   call_window_records[...]                               their record counts
 *   records[record_offsets[m] ..)                          the records {position, deviation} (8 bytes)
 * with window and record indices relative to the first block of the block's sample (sample_first[m]: index of that block; the
 * blocks of a sample are consecutive). block_flags[4 * m + 3] != 0: the block is not parsed (not asked for, refused by the
 * inflater, or irregular) -- parse the sample on the host. The inflated bytes are copied back (*out) only if some block was left
 * unparsed (bytes_copied). Arrays are page-locked, callee-owned, valid until the next run on this inflater. */
typedef struct pd_parsed_blocks
{
    const void     *profile_windows;
    const void     *call_windows;
    const uint32_t *call_window_records;
    const void     *records;
    const uint64_t *profile_window_offsets, *call_window_offsets, *record_offsets;     /* n_members + 1 each */
    const uint32_t *block_flags;            /* 4 words per block: 256-bp windows, 30-bp windows, records, flags */
    uint32_t        bytes_copied;
} pd_parsed_blocks;
int pd_inflater_run_parsed(pd_inflater *inflater, const uint64_t *member_offsets, uint32_t n_members, const uint64_t *out_offsets,
                           const uint32_t *sample_first, const uint8_t *parse_member, const uint8_t **out, const uint32_t **status,
                           const uint64_t **out_lengths, pd_parsed_blocks *parsed, double *kernel_ms);

/* ... and the parsed records SUMMARISED block by block for the loader's tables (ABI 14): what filing them needs to know when the
 * maxLoad rule cannot bite (reference: ChromosomeProfile::add, profile_structure_popdel_call.h:1084-1112, asks
 * CyclicEndEntryTable::getEndCount, :870-924, after every record how many pairs are over -- a function of how many pairs BEGIN and
 * how many END in each 30-bp window as long as no record is dropped and the counting cursor stays in its slot).
 * pd_inflater_run_summarized() is pd_inflater_run_parsed() with, for every parsed block m (a pair's end position being position +
 * max(0, deviation + inner_offset[m]); inner_offset[m] = median - 2 * read length of the block's read group; options->starts_room /
 * ends_room windows of counts per block),
 *   block_summaries[m]                      number of records (n; 0: nothing else is filled in), first and last start position, the
 *                                           window (position / 30) of the counts' first entry (= that of the first position), the
 *                                           number of windows up to the last end window, smallest / largest / first / last end window,
 *                                           flags: 1 start positions in non-decreasing order, 2 last windows in non-decreasing order,
 *                                           4 a count reached 255 (saturated), 8 a window outside the room (counts unusable)
 *   block_counts[m * (starts_room + ends_room) + b]                records that BEGIN in window bin_base + b (b < starts_room; one
 *                                                                  byte, saturating)
 *   block_counts[m * (starts_room + ends_room) + starts_room + b]  records that END there (b < ends_room)
 * in the same page-locked, callee-owned arrays' lifetime as the rest of pd_parsed_blocks. A block spans one index region (10,000 bp
 * by default: 335 windows of starts, ends a little further). The host's ChromProfile::addBulk() files whole blocks from this without
 * reading their records; summarizeRun() (chrom_profile.cpp) is the same computation on the host, the kernel's yardstick
 * (tests/test_gpu_summarize.py).
 * options->lean != 0: the 30-bp windows are not written out. profile_windows then holds per 256-bp window {chrom, beginPos, first record,
 * number of records, position of the first record, position of the last} (24 bytes; record indices relative to the sample's first
 * block), call_windows and call_window_records are NULL: the 30-bp windows of a 256-bp window are the runs of equal position / 30
 * among its records (convertWindow anchors the grid at a multiple of 30), which a host reads off the records where it needs them --
 * at the two ends of what a round files -- and a third less is copied back. */
typedef struct pd_block_summary
{
    uint32_t n, first_pos, last_pos, bin_base, n_bins, min_end_bin, max_end_bin, first_end_bin, last_end_bin, flags;
} pd_block_summary;
typedef struct pd_block_summaries
{
    const pd_block_summary *block_summaries;    /* n_members */
    const uint8_t          *block_counts;       /* n_members x (starts_room + ends_room) */
    uint32_t                starts_room, ends_room;
    const uint64_t         *record_index;       /* n_members: see keep_below / keep_from (NULL if all records were copied back) */
} pd_block_summaries;
/* keep_below / keep_from (both or neither; per member): with them the RECORDS of a block are copied back only if one of them begins
 * below keep_below[m] or at / beyond keep_from[m] -- a loader that files whole blocks from their summaries reads records only at the
 * two ends of what a round files, and the window loop takes its entries from the device's copy (pd_walk_source): four fifths of
 * the records never cross the link. parsed->records then holds the kept blocks' records packed in block order, record_index[m] says
 * where block m's begin (UINT64_MAX: not copied back; kept blocks that follow each other in a sample follow each other there),
 * and parsed->record_offsets keeps counting ALL records (what indices into the device's array and the lean windows refer to). */
typedef struct pd_summarize_options
{
    const int32_t  *inner_offset;       /* n_members */
    uint32_t        starts_room, ends_room;
    uint32_t        lean;
    const uint32_t *keep_below, *keep_from;
} pd_summarize_options;
/* How many parsed chunks the inflater has run so far (the number of the one its device buffers hold; pd_walk_source::run_id). */
uint64_t pd_inflater_run_id(pd_inflater *inflater);
int pd_inflater_run_summarized(pd_inflater *inflater, const uint64_t *member_offsets, uint32_t n_members, const uint64_t *out_offsets,
                               const uint32_t *sample_first, const uint8_t *parse_member, const pd_summarize_options *options,
                               const uint8_t **out, const uint32_t **status, const uint64_t **out_lengths, pd_parsed_blocks *parsed,
                               pd_block_summaries *summaries, double *kernel_ms);

/* Text of the last error on this context (or of the last failed pd_ctx_create if ctx == NULL). */
const char *pd_last_error(const pd_ctx *ctx);

/* Start the GPU runtime on `device` ahead of pd_ctx_create() (loading the code objects and creating the device context
 * takes 0.1-0.2 s): a host that has other start-up work -- `popdel call` reads its profile headers -- calls this from a
 * second thread first. No reference counterpart. Returns 0 on success. */
int pd_device_warmup(int device);

/* Name of the implementation behind the ABI ("hip-gfx950" for the product library). */
const char *pd_backend_name(void);

int pd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* POPDEL_HIP_H_ */
