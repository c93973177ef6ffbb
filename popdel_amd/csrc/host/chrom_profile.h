// Window state machine of `popdel call`: per read group a three-slot cyclic table of read-pair start entries,
// a three-slot cyclic table of end entries, and the set of read pairs that are "active" in the current window.
// This defines WHICH pairs the genotyping kernels see in WHICH window, including the loader's behaviour around
// the 200 kb slot borders, so it is restated operation for operation.
// Behavioural reference: popdel_call/profile_structure_popdel_call.h:173-450 (CyclicStartEntryTable),
// :671-948 (CyclicEndEntryTable), :1035-1785 (ChromosomeProfile), :1878-1927 (switch helpers).
//
// Instead of exposing the active set to a per-window callee, this class records for every pair the window
// index at which it enters and leaves the active set ("pair instances"); that event log is what is shipped to
// the GPU (include/popdel_hip.h, pd_segment).
#ifndef PDH_CHROM_PROFILE_H_
#define PDH_CHROM_PROFILE_H_

#include <algorithm>
#include <atomic>
#include <functional>
#include <cstdint>
#include <cstddef>
#include <limits>
#include <unordered_map>
#include <vector>

#include "huge_pool.h"

namespace pdh {

struct StartEntry
{
    uint32_t startPos;
    int32_t  deviation;
};

struct EndEntry
{
    uint32_t id;
    uint32_t lastWindow;
};

// Three slots used round-robin: one is written while the previous one is read.
// (Cache-line aligned, like EndSlots: neighbouring read groups are filed by different threads, and every record that is
// filed writes to the object -- vector ends, counters.)
class alignas(64) StartSlots
{
public:
    // A run of entries that is NOT held here (round 6): the records a device parsed and summarised, filed by
    // ChromProfile::addBulk() without a copy -- the window loop on the device takes them from the loader's records where they lie
    // (pd_walk_source), and this table needs of them what `total`, `frontPos`, `backPos` and `sorted` say. Whoever does need the
    // entries themselves (the host's window loop taking over, end entries written out) has them read again from the profile:
    // ChromProfile::materializeStarts(), through the filer's `refill` callback.
    struct VirtualRun
    {
        size_t   explicitBefore;    // entries.size() when the run was filed: its place in filing order
        uint32_t n;                 // entries
        // where a reader of the profile finds them again: the 30-bp window (position) at which the stretch of records they were
        // filed with begins -- the first record at or beyond it is that stretch's first, positions being in order --, how many
        // records of the stretch precede them, the position the round's reader had jumped to (it sees what a jump there sees:
        // a 256-bp window that begins in the index region before reaches over the border), the filer's own note (CallRun: which contig)
        uint32_t firstWindow, skip, jumpPos, tag;
    };
    struct Slot
    {
        HugeVector<StartEntry> entries;            // (without the virtual runs)
        std::vector<VirtualRun> virtualRuns;
        size_t   total = 0;        // entries filed: entries.size() + the virtual runs'
        unsigned frontPos = 0, backPos = 0;        // start positions of the first and the last entry filed (total != 0)
        unsigned right = 0;        // exclusive right border
        unsigned idOffset = 0;     // id of entries[0]
        unsigned endBorder = 0;    // right border of the END slot the entries' end records were filed under (ChromProfile::add)
        bool readable = false;
        bool sorted = true;        // entries arrived in non-decreasing start order (what the closed-form walk relies on)
        void reset(unsigned idO, unsigned r)
        {
            entries.clear();
            virtualRuns.clear();
            total = 0;
            frontPos = backPos = 0;
            right = r;
            idOffset = idO;
            readable = false;
            sorted = true;
        }
        bool isVirtual() const { return !virtualRuns.empty(); }
        // an entry is filed (any way): what the slot remembers about its entries beyond the entries
        void note(unsigned firstPos, unsigned lastPos, size_t n)
        {
            if (total == 0)
                frontPos = firstPos;
            else if (firstPos < backPos)
                sorted = false;
            backPos = lastPos;
            total += n;
        }
    };

    explicit StartSlots(unsigned firstWin = 0, unsigned windows = 10000);

    bool needsSwitch(unsigned firstWindow) const { return firstWindow >= slots[writeSlot].right; }
    bool tooBigForNext(unsigned firstWindow) const { return firstWindow >= slots[writeSlot].right + numWindows; }
    void switchWrite();
    bool switchRead();
    bool switchBoth(unsigned & rightBorderOut);
    void add(unsigned startPos, int deviation);

    bool readAtEnd() const { return readIdx >= slots[readSlot].total; }
    bool goNext() { ++readIdx; return !readAtEnd(); }
    const StartEntry & current() const { return slots[readSlot].entries[readIdx]; }     // (the read slot holds all its entries: host's window loop)
    unsigned currentId() const { return static_cast<unsigned>(readIdx) + slots[readSlot].idOffset; }
    // The read cursor as the closed-form plan moves it (entries of the read slot in start order): behind every entry that starts at
    // or before `pos`, never backwards. With virtual runs in the read slot the entries cannot be counted: the cursor is then kept
    // as that position (readByPos / readPos; readIdx stays a lower bound and is exact at 0 and at the end -- all the plan asks),
    // and becomes a number again when the slot is materialised.
    void cursorToStart()
    {
        readIdx = 0;
        readByPos = false;
    }
    void cursorUpTo(unsigned pos);
    unsigned firstStart() const { return slots[readSlot].frontPos; }        // start position of the slot's first entry (total != 0)
    bool     readByPos = false;
    unsigned readPos = 0;
    unsigned slotOfId(unsigned id) const;
    const StartEntry & entryOfId(unsigned id) const;

    Slot     slots[3];
    size_t   readIdx = 0;          // position of the read cursor inside slots[readSlot]
    unsigned readSlot = 2;
    unsigned writeSlot = 0;
    unsigned insertionCount = 0;
    unsigned numWindows;
};

// Two ways to keep the end tables (round 4). HEAVY: every slot's entries ordered by last window at all times, a new record
// inserted behind the last entry that does not end later (the reference's upper_bound insert) -- what the window loop on the
// host walks. LIGHT: while the window loop runs in closed form on the device nobody walks these entries; what the loader
// itself needs of them is how many end before a position (the maxLoad rule's getEndCount(), the plan's cursors) and the
// last window of a slot. A light slot therefore keeps its records in FILING order (an append) plus a count per 30-bp window,
// and answers the counting questions from the counts -- entry for entry what the ordered array would answer, because an
// entry's index in the ordered array is the number of entries that end before it (ties in filing order). Going back to
// heavy orders the log once (stable by last window = the order the inserts would have produced), so the host's window loop
// can take over at any segment.
class alignas(64) EndSlots
{
public:
    // Light slots need not write their entries out at all (round 6): a long run of records filed at once (ChromProfile::
    // addBulk) is remembered as WHERE its start entries lie -- which start slot of the read group, from which entry on, how
    // many, the id of the first, the inner offset that turns {position, deviation} into an end position -- and becomes
    // EndEntry records only if somebody asks for them (leaving light mode; a record outside the counts' range). A start slot
    // that is about to be recycled under such a reference has it written out first (ChromProfile::settleDeferred).
    struct Deferred
    {
        size_t   explicitBefore;    // entries.size() when the run was filed: its place in filing order
        uint32_t startSlot;         // the start slot (of the same read group) that holds its records
        uint32_t first, n;          // start entries [first, first + n) of that slot
        uint32_t firstId;           // id of the first of them (ids are consecutive)
        int32_t  innerOffset;       // end position = start + max(0, deviation + innerOffset)
    };
    struct Slot
    {
        HugeVector<EndEntry> entries;    // heavy: ordered by lastWindow (stable); light: in filing order, `ordered` says if that is the same
                                         // (light: without the runs in `deferred`)
        std::vector<Deferred> deferred;  // light: runs filed in bulk, not written out
        size_t total = 0;                // entries filed (entries.size() + the deferred runs' records), both modes
        unsigned backWin = 0;            // last window of the entry filed last (meaningful while total != 0; light)
        unsigned right = 0;
        bool readable = false;
        // light:
        HugeVector<uint16_t> hist;       // entries per 30-bp window, bin b = window (baseBin + b) * 30
        unsigned baseBin = 0;
        unsigned maxWin = 0;             // largest lastWindow filed (meaningful while there are entries)
        unsigned lowBin = 0, highBin = 0;    // bins in use: [lowBin, highBin)
        bool ordered = true;
        unsigned peak = 0;               // the largest count in hist (kept exact: every increment looks at it)
        unsigned curBin[2] = {0, 0}, curCum[2] = {0, 0};    // prefix cursors over hist: filing side, reading side
        void reset(unsigned r)
        {
            entries.clear();
            deferred.clear();
            total = 0;
            backWin = 0;
            right = r;
            readable = false;
            for (unsigned b = lowBin; b < highBin && b < hist.size(); ++b)
                hist[b] = 0;
            lowBin = highBin = 0;
            maxWin = 0;
            peak = 0;
            ordered = true;
            curBin[0] = curBin[1] = curCum[0] = curCum[1] = 0;
        }
    };

    explicit EndSlots(unsigned firstWin = 0, unsigned windows = 10000);

    bool needsSwitch(unsigned lastWindow) const { return lastWindow >= slots[writeSlot].right; }
    void switchWrite();
    void fixConsecutiveSwitch(unsigned & activeLoad);
    bool switchRead();
    void add(unsigned id, unsigned endPos);
    unsigned countEnded(unsigned pos);      // getEndCount() of the reference

    bool readAtEnd() const { return readIdx >= slots[readSlot].total; }
    bool goNext() { ++readIdx; return !readAtEnd(); }
    const EndEntry & current() const { return slots[readSlot].entries[readIdx]; }   // heavy only

    // the read cursor behind every entry of the read slot that is over at `pos` (lastWindow + shift < pos), never backwards
    void advanceReadTo(unsigned pos, unsigned shift);
    unsigned lastWindowOfReadSlot() const;      // of its last entry in lastWindow order (the slot must not be empty)
    void setLight(bool on);
    bool light = false;

    // add() / countEnded() with the everyday case of light tables spelled out inline (a record filed into a slot that has
    // entries, inside the windows its counts cover; a question that moves the counting cursor a window or two forward and
    // leaves the end-position cursor inside its slot); everything else goes through the general functions. Same state
    // afterwards, same answers.
    void addFast(unsigned id, unsigned endPos)
    {
        const unsigned bin = endPos / 30u, lastWin = bin * 30u;
        Slot & slot = slots[lastWin >= slots[writeSlot].right ? (writeSlot + 1) % 3 : writeSlot];
        const unsigned b = bin - slot.baseBin;              // (wraps below the base: out of range)
        if (light && slot.total != 0 && b < slot.hist.size() && slot.hist[b] != 0xffffu)
        {
            slot.ordered = slot.ordered && lastWin >= slot.backWin;
            slot.backWin = lastWin;
            ++slot.total;
            if (b < slot.lowBin)
                slot.lowBin = b;
            if (b >= slot.highBin)
                slot.highBin = b + 1;
            if (lastWin > slot.maxWin)
                slot.maxWin = lastWin;
            const unsigned count = ++slot.hist[b];
            if (count > slot.peak)
                slot.peak = count;
            slot.curCum[0] += b < slot.curBin[0] ? 1u : 0u;
            slot.curCum[1] += b < slot.curBin[1] ? 1u : 0u;
            slot.entries.push_back(EndEntry{id, lastWin});
            return;
        }
        add(id, endPos);
    }
    // How many of `n` records (n >= 1, start positions in non-decreasing order from firstPos to lastPos) may be filed without
    // asking countEnded() in between: n (or fewer) if the answers to the questions left out are certain to come from the
    // everyday case of countEndedFast() -- light tables, positions from the counting cursor on and inside the windows the
    // counts cover, and the cursor unable to reach the end of its slot: either every record of the block is filed into that
    // very slot (intoWriteSlot, and the cursor is in the write slot: the record filed last is never over when the question
    // behind it is asked) or the slot holds more entries than could be over even if all of the block were (the cases in which
    // getEndCount() moves on to the next slot answer question by question what one question would not) -- so that one
    // question behind the block gives the sum of their answers and leaves the same state. Else 0.
    size_t quietBlock(unsigned firstPos, unsigned lastPos, size_t n, bool intoWriteSlot) const
    {
        const Slot & slot = slots[endPosSlot];
        const size_t have = slot.total;
        if (!light || dNextOffset != 0 || dNext >= have)
            return 0;
        const unsigned firstBins = firstPos / 30u, lastBins = lastPos / 30u;
        const unsigned tFirst = firstBins > slot.baseBin ? firstBins - slot.baseBin : 0u;
        const unsigned tLast = lastBins > slot.baseBin ? lastBins - slot.baseBin : 0u;
        if (tFirst < slot.curBin[0] || tLast > slot.hist.size())
            return 0;
        if (intoWriteSlot && endPosSlot == writeSlot)
            return n;
        size_t over = slot.curCum[0];
        const unsigned from = slot.curBin[0] > slot.lowBin ? slot.curBin[0] : slot.lowBin;
        const unsigned to = tLast < slot.highBin ? tLast : slot.highBin;
        for (unsigned b = from; b < to; ++b)
            over += slot.hist[b];
        // (fewer records reach no further: what holds for n holds for any shorter block)
        if (over + 1 >= have)
            return 0;
        return std::min<size_t>(n, have - over - 1);
    }
    // addFast() for records id, id + 1, ... whose last windows (30 * bins[j]) all lie below the write slot's border. The everyday
    // case -- the slot has entries, all bins inside the counted range, none of their counts near saturation -- is taken for the
    // whole block at once: one pass for what the slot remembers about its entries (extremes, order), one for the counts, the
    // entries appended in one copy. Anything else: record by record through addFast().
    void addBlockToWriteSlot(unsigned id, const unsigned * bins, size_t n)
    {
        constexpr size_t kMost = 256;
        Slot & slot = slots[writeSlot];
        if (light && slot.total != 0 && n <= kMost)
        {
            EndEntry fresh[kMost];
            unsigned minBin = 0xffffffffu, maxBin = 0, before = slot.backWin;
            bool ordered = true;
            for (size_t j = 0; j < n; ++j)
            {
                const unsigned bin = bins[j], lastWin = bin * 30u;
                fresh[j] = EndEntry{id + static_cast<unsigned>(j), lastWin};
                minBin = bin < minBin ? bin : minBin;
                maxBin = bin > maxBin ? bin : maxBin;
                ordered = ordered && lastWin >= before;
                before = lastWin;
            }
            const unsigned base = slot.baseBin;
            if (minBin >= base && maxBin - base < slot.hist.size())
            {
                uint16_t * hist = slot.hist.data();
                // (no count of the slot near saturation: `peak` is the largest of them -- a pass over the block's own counts
                // used to ask this, a tenth of the filing's time)
                if (slot.peak < 0xffffu - kMost)
                {
                    const unsigned cur0 = slot.curBin[0], cur1 = slot.curBin[1];
                    unsigned cum0 = 0, cum1 = 0, peak = slot.peak;
                    for (size_t j = 0; j < n; ++j)
                    {
                        const unsigned b = bins[j] - base;
                        const unsigned count = ++hist[b];
                        peak = count > peak ? count : peak;
                        cum0 += b < cur0 ? 1u : 0u;
                        cum1 += b < cur1 ? 1u : 0u;
                    }
                    slot.peak = peak;
                    slot.curCum[0] += cum0;
                    slot.curCum[1] += cum1;
                    slot.ordered = slot.ordered && ordered;
                    if (minBin - base < slot.lowBin)
                        slot.lowBin = minBin - base;
                    if (maxBin - base >= slot.highBin)
                        slot.highBin = maxBin - base + 1;
                    if (maxBin * 30u > slot.maxWin)
                        slot.maxWin = maxBin * 30u;
                    slot.entries.insert(slot.entries.end(), fresh, fresh + n);
                    slot.total += n;
                    slot.backWin = before;
                    return;
                }
            }
        }
        for (size_t j = 0; j < n; ++j)
            addFast(id + static_cast<unsigned>(j), bins[j] * 30u);     // (any end position inside the window files the same)
    }
    unsigned countEndedFast(unsigned pos)
    {
        Slot & slot = slots[endPosSlot];
        const unsigned n = static_cast<unsigned>(slot.total);
        if (light && dNextOffset == 0 && dNext < n)
        {
            const unsigned bins = pos / 30u;                // windows 0 .. bins-1 end before the window of pos
            const unsigned t = bins > slot.baseBin ? bins - slot.baseBin : 0u;
            if (t >= slot.curBin[0] && t <= slot.hist.size())
            {
                unsigned cc = slot.curCum[0];
                const unsigned from = slot.curBin[0] > slot.lowBin ? slot.curBin[0] : slot.lowBin;
                const unsigned to = t < slot.highBin ? t : slot.highBin;
                for (unsigned b = from; b < to; ++b)
                    cc += slot.hist[b];
                slot.curBin[0] = t;
                slot.curCum[0] = cc;
                if (cc <= dNext)
                    return 0;
                const unsigned k = cc - dNext;              // (cc <= n: it counts entries of this slot)
                if (dNext + k < n)
                {
                    dNext += k;
                    return k;
                }
            }
        }
        return countEnded(pos);
    }

    Slot     slots[3];
    size_t   readIdx = 0;
    unsigned dNext = 0;
    unsigned dNextOffset = 0;
    unsigned writeSlot = 0;
    unsigned endPosSlot = 0;
    unsigned readSlot = 2;
    unsigned numWindows;

    // ---- runs filed in bulk (Deferred) ------------------------------------------------------------------------------------
    // the start table of the same read group (ChromProfile sets it): where the deferred runs' records lie; a start slot with
    // virtual runs is materialised first (owner / rgIndex)
    const StartSlots * startsOf = nullptr;
    class ChromProfile * owner = nullptr;
    unsigned rgIndex = 0;
    // every deferred run of `slot` written out as EndEntry records, in filing order
    void settle(Slot & slot);
    // ... those of all slots that refer to start slot `startSlot` (it is about to be recycled), except in end slot `spare`
    // (3: none), which is about to be reset itself
    void settleRefsTo(unsigned startSlot, unsigned spare);
    bool hasDeferred() const { return !slots[0].deferred.empty() || !slots[1].deferred.empty() || !slots[2].deferred.empty(); }

private:
    void addLight(Slot & slot, unsigned id, unsigned lastWin);
    unsigned countEndedLight(unsigned pos);
    unsigned lessCount(Slot & slot, unsigned pos, int side);      // entries of the slot with lastWindow < pos
    void prepareHist(Slot & slot);
    void leaveLight();
};

// ---- records summarised (round 6: filing without looking at the records) ---------------------------------------------------
// What ChromProfile::addBulk() needs to know about a run of records -- {position, deviation} pairs of one read group in the order
// the reader delivers them -- instead of the records: how many begin and how many END in each 30-bp window (window binBase + b),
// the extremes, and whether positions and last windows come in non-decreasing order. The device computes it per profile block
// where the records were parsed (inflate.hip, pd_inflater_run_summarized: one byte per window, 255 = saturated); the driver adds
// the blocks of a run up (CallRun::fileRun); summarizeRun() is the kernel's computation on the host (tests, its yardstick).
struct RunSummary
{
    uint32_t n = 0;                     // records covered
    uint32_t firstPos = 0, lastPos = 0; // start positions of the first and the last of them
    uint32_t binBase = 0;               // window (position / 30) of the counts' first entry = window of firstPos
    uint32_t nBins = 0;                 // windows up to the last end window: binBase + nBins - 1
    uint32_t minEndBin = 0, maxEndBin = 0, firstEndBin = 0, lastEndBin = 0;     // absolute window numbers
    uint32_t flags = 0;
    static constexpr uint32_t kStartsInOrder = 1u, kEndsInOrder = 2u, kSaturated = 4u, kNoRoom = 8u;
};
// records {pos, dev} x nRecords, end position = pos + max(0, dev + innerOffset); covered is the longest prefix whose pairs all
// have their last window below `borderBin`; starts[] has room for `roomStarts` windows, ends[] for `roomEnds`
void summarizeRun(const uint32_t * posDev, size_t nRecords, int innerOffset, uint32_t borderBin, uint32_t roomStarts, uint32_t roomEnds,
                  RunSummary & out, uint8_t * starts, uint8_t * ends);

// ---- the window loop in closed form -----------------------------------------------------------------------------------
// What the loop of processSegment() -- one extra step of nextWindow(), initializeActive(), then the walk -- does to a read
// group is a function of a few numbers (WalkShape) and, per pair, of its start, its last window and the slot its end record
// sits in. The GPU backend evaluates that per pair itself (pd_walk_submit, include/popdel_hip.h); the host then only plans
// the segment (ChromProfile::planSegment: O(1) per read group from the heads and tails of the slot tables) instead of
// walking it. WalkReference is the same evaluation on the host: the checker of the plan (POPDEL_WALK_CHECK) and of the kernel.
struct WalkShape
{
    unsigned firstPos = 0;     // position of window 0
    unsigned stepPos = 0;      // position of the loader's step ahead of window 0 (hasStep)
    unsigned nWindows = 0;
    bool     hasStep = false;
};

// A pair as seen by the window loop of one segment: active for window indices [on, off).
struct PairInstance
{
    int32_t  dev;
    uint32_t start;
    uint32_t on;
    uint32_t off;
    uint32_t id;           // the reference's id of the pair (entry of the cyclic start table); not part of pd_segment
};

// One mutation of a read group's active set, in the order the reference performs them (emplace / erase on its
// std::unordered_set, profile_structure_popdel_call.h:1143-1207). Recorded on request only (tests: the oracle replays them
// on a real std::unordered_set to visit the pairs in the reference's order). `window`: index of the first window of the
// segment that is genotyped after the mutation (0 for mutations the loader makes between segments).
struct ActiveOp
{
    uint32_t window;
    uint32_t id;
    uint32_t add;
};

// (what a virtual run notes of where its records are found again: StartSlots::VirtualRun)
struct VirtualNote
{
    uint32_t firstWindow = 0, skip = 0, jumpPos = 0, tag = 0;
};

class ChromProfile
{
public:
    ChromProfile(unsigned numReadGroups, const std::vector<unsigned> & maximumLoad, unsigned bufferSize);

    bool isHighCov(unsigned rg) const { return active[rg].size() >= maxLoad[rg]; }
    void add(unsigned rg, unsigned startPos, unsigned endPos, int deviation);
    // add() for a run of records of one read group ({pos, dev} pairs as the reader delivers them; the pair's end is
    // pos + max(0, dev + innerOffset), innerOffset = median - 2 * read length): the same operations in the same order
    void addRun(unsigned rg, const uint32_t * posDev, size_t n, int innerOffset);
    // addRun() for records [0, s.n) WITHOUT reading them (posDev: they are only copied into the start slot; nullptr: not even that --
    // the start slot notes a virtual run, StartSlots::VirtualRun, with `note` saying where its refill finds the records): from the summary the tables
    // can tell whether filing them record by record would have been the everyday case throughout -- light tables, the end
    // table's counting cursor at home in the write slot, every pair ending below that slot's border and inside the windows its
    // counts cover, positions in order, and the load below the cap behind EVERY record (followed window by window from the
    // counts: exact, not a bound) -- and if so files them at once: the same tables, cursors and load as addRun() leaves, the end
    // entries remembered as a reference to the start entries (EndSlots::Deferred). Returns false, with nothing changed, if any of
    // it does not hold: the caller then files the records with addRun().
    bool addBulk(unsigned rg, const uint32_t * posDev, const RunSummary & s, const uint16_t * startCounts, const uint16_t * endCounts,
                 int innerOffset, const VirtualNote & note = VirtualNote());
    // The entries of virtual runs, for whoever needs them: `refill(rg, run, dst)` writes run.n entries -- those of the records that
    // were filed, in order -- to dst (the filer reads them again from the profile). materializeStarts() makes a slot hold all its
    // entries again (and its read cursor a number); materializeAllStarts(): every slot of every read group.
    std::function<void(unsigned rg, const StartSlots::VirtualRun & run, StartEntry * dst)> refill;
    void materializeStarts(unsigned rg, unsigned slot);
    void materializeAllStarts();
    uint64_t materialized() const { return materialized_; }
    // room for n more records in the slots being written (one allocation instead of the doublings of push_back)
    void reserveRun(unsigned rg, size_t n);
    // ... for a run that will mostly be filed from its summary (addBulk): the start slot takes every record, the end slot a few
    void reserveStarts(unsigned rg, size_t n);
    void initializeActive();
    bool nextWindow(unsigned shift);
    void resetTo(unsigned pos);
    void fullReset();
    // The switch helpers write the right border of the new read slot to *border (default: currentRightBorder); parallel
    // callers pass a private variable and publish it afterwards in sample order.
    void performSwitches(const std::vector<unsigned> & rgs, unsigned * border = nullptr);
    void performPartialSwitches(const std::vector<unsigned> & rgs, unsigned * border = nullptr);
    bool allEmpty() const;
    // light end tables (see EndSlots) while nobody walks them: the driver turns this on with the device's window loop
    void setLightEnds(bool on);
    bool lightEnds() const { return lightEnds_; }
    // while someone else evaluates the walk (and keeps the active pairs): how many pairs are active there
    std::function<size_t()> externalActive;

    // --- event log (segment builder) -------------------------------------------------------------------
    // Call before the first genotyped window of a segment: carried-over pairs become instances with on = 0.
    void beginSegmentLog();
    // Call after each genotyped window (before nextWindow()).
    void windowDone() { ++logWindow_; }
    // The whole window loop of one segment in one call: equivalent to
    //     while (!profilesAtEnd) { windowDone(); profilesAtEnd = !nextWindow(shift); }
    // but read groups are walked independently (and in parallel): within a step they only share the position, and the
    // loop ends after min(steps to the slot border, steps until the last read group runs dry). Returns the number of
    // genotyped windows.
    unsigned walkSegment(unsigned shift);
    // Close all open instances at `off = number of windows` and return the per-read-group instance lists.
    // Still-active pairs stay in the active set for the next segment.
    std::vector<std::vector<PairInstance>> & endSegmentLog();
    unsigned loggedWindows() const { return logWindow_; }
    // mutation log (see ActiveOp): everything since the last takeActiveOps()
    void enableActiveOpLog() { opLogging_ = true; ops_.resize(active.size()); }
    std::vector<std::vector<ActiveOp>> & activeOps() { return ops_; }
    void clearActiveOps() { for (auto & v : ops_) v.clear(); }

    // --- closed-form walk (see WalkShape) -----------------------------------------------------------------------------
    // Can the coming segment be planned instead of walked? (Not: a segment without windows whose set-up still activates
    // pairs, unsorted start entries.) Call before planSegment().
    bool canPlanSegment(unsigned shift) const;
    // Everything processSegment() does to the positions and cursors, without touching a pair: returns the segment's
    // shape and fills per read group the first-segment rule (equalRun) and the slot switches since the last plan.
    WalkShape planSegment(unsigned shift, std::vector<uint8_t> & equalRun, std::vector<uint8_t> & switches);
    // (after planSegment: forget the counted switches; the closed-form state of whoever evaluates the walk is in step)
    const std::vector<uint8_t> & switchCounts() const { return switches_; }
    // (check mode: plan, note the result, put everything planSegment() touches back, then walk)
    struct PlanState
    {
        unsigned currentPos, globalMinPos, logWindow;
        bool profilesAtEnd, activeKnownEmpty;
        std::vector<uint8_t> switches;
        std::vector<std::pair<size_t, size_t>> cursors;
    };
    // after a segment was WALKED (not planned): the bookkeeping planSegment() does for the next plan
    void noteSegmentWalked() { std::fill(switches_.begin(), switches_.end(), static_cast<uint8_t>(0)); activeKnownEmpty_ = false; }
    PlanState savePlanState() const;
    void restorePlanState(const PlanState & s);
    bool activeKnownEmpty() const { return activeKnownEmpty_; }
    // a pair the evaluator of the closed-form walk holds as active, back into the tables' active set (leaving that mode)
    void materializeActive(unsigned rg, uint32_t id, uint32_t start, int32_t dev) { active[rg].emplace(id, ActiveRec{dev, start, kNoLog}); }
    void setActiveKnownEmpty(bool v) { activeKnownEmpty_ = v; }

    std::vector<StartSlots> starts;
    std::vector<EndSlots> ends;
    // Active read pairs of one read group: id -> what the window loop needs to know about the pair.
    struct ActiveRec
    {
        int32_t  dev;
        uint32_t start;
        uint32_t logIdx;      // index of the open instance in the segment log (kNoLog if none)
    };
    // id -> ActiveRec in one flat array (open addressing, linear probing, deletion by backward shift): the reference's
    // std::unordered_set costs one heap node per activation, which on 64 host threads means allocator traffic for every
    // read pair. Only membership matters; whoever needs an order sorts by id.
    class ActiveSet
    {
    public:
        ActiveSet() : slots_(64), mask_(63) {}
        size_t size() const { return size_; }
        bool empty() const { return size_ == 0; }
        ActiveRec * find(uint32_t id)
        {
            for (size_t i = hash(id) & mask_;; i = (i + 1) & mask_)
            {
                Slot & s = slots_[i];
                if (!s.used)
                    return nullptr;
                if (s.id == id)
                    return &s.rec;
            }
        }
        void emplace(uint32_t id, const ActiveRec & rec)          // id must not be present
        {
            if ((size_ + 1) * 2 > slots_.size())
                grow();
            insert(id, rec);
            ++size_;
        }
        void erase(uint32_t id)
        {
            size_t i = hash(id) & mask_;
            for (;; i = (i + 1) & mask_)
            {
                if (!slots_[i].used)
                    return;
                if (slots_[i].id == id)
                    break;
            }
            // backward shift: close the gap so that probing sequences stay intact
            size_t hole = i;
            for (size_t j = (i + 1) & mask_; slots_[j].used; j = (j + 1) & mask_)
            {
                const size_t home = hash(slots_[j].id) & mask_;
                const bool movable = ((j - home) & mask_) >= ((j - hole) & mask_);
                if (movable)
                {
                    slots_[hole] = slots_[j];
                    hole = j;
                }
            }
            slots_[hole].used = false;
            --size_;
        }
        template <typename F>
        void forEach(F f)
        {
            for (Slot & s : slots_)
                if (s.used)
                    f(s.id, s.rec);
        }

    private:
        struct Slot
        {
            uint32_t  id = 0;
            bool      used = false;
            ActiveRec rec{};
        };
        static size_t hash(uint32_t id) { return static_cast<size_t>(id) * 0x9E3779B1u; }
        void insert(uint32_t id, const ActiveRec & rec)
        {
            size_t i = hash(id) & mask_;
            while (slots_[i].used)
                i = (i + 1) & mask_;
            slots_[i].id = id;
            slots_[i].used = true;
            slots_[i].rec = rec;
        }
        void grow()
        {
            std::vector<Slot> old;
            old.swap(slots_);
            slots_.assign(old.size() * 2, Slot());
            mask_ = slots_.size() - 1;
            for (const Slot & s : old)
                if (s.used)
                    insert(s.id, s.rec);
        }
        std::vector<Slot> slots_;
        size_t mask_;
        size_t size_ = 0;
    };
    std::vector<ActiveSet> active;
    unsigned globalMinPos = std::numeric_limits<unsigned>::max();
    unsigned currentPos = std::numeric_limits<unsigned>::max();
    bool     profilesAtEnd = false;
    unsigned currentRightBorder;
    std::vector<unsigned> activeLoad;
    std::vector<unsigned> maxLoad;

    static constexpr uint32_t kNoLog = 0xffffffffu;

private:
    void activate(unsigned rg, unsigned id, unsigned logWindow);
    void deactivate(unsigned rg, unsigned id, unsigned logWindow);
    void nextStartWindow(unsigned rg, unsigned logWindow);
    bool nextEndWindow(unsigned rg, unsigned logWindow);
    void drainEndReadSlot(unsigned rg);
    unsigned walkReadGroup(unsigned rg, unsigned pos0, unsigned shift, unsigned maxSteps);

    std::atomic<uint64_t> materialized_{0};       // entries of virtual runs read back (refill)
    std::vector<uint8_t> switches_;        // per read group: slot switches since the last planned / walked segment (2 = two or more)
    bool activeKnownEmpty_ = true;         // no read group has an active pair (start of a run; closed-form mode keeps no active sets)
    bool lightEnds_ = false;
    bool logging_ = false;
    bool opLogging_ = false;
    std::vector<std::vector<ActiveOp>> ops_;
    unsigned logWindow_ = 0;
    std::vector<std::vector<PairInstance>> log_;
};

// The walk of one read group in closed form, with the state it carries from segment to segment: the pairs still active
// (each knows whether its end record sits in the read slot or the one after) and the start entries of the read slot that
// no window has reached yet. Host reference of the device kernel (walk.hip).
class WalkReference
{
public:
    struct Pair
    {
        uint32_t start;
        int32_t  dev;
        uint32_t id;
        uint32_t lastWin;
        uint8_t  next;         // its end record sits in the slot after the end read slot
    };
    explicit WalkReference(unsigned nReadGroups) : active_(nReadGroups), pending_(nReadGroups) {}
    // One read group, one segment. `entries`: the new start slot (only looked at if switches != 0). Appends the instances
    // in the order the walk logs them (carried pairs by id, then activations in entry order).
    void segment(unsigned rg, const WalkShape & shape, unsigned shift, unsigned switches, bool equalRun,
                 const StartSlots::Slot * slot, unsigned median, unsigned readLength, std::vector<PairInstance> & out);
    const std::vector<Pair> & active(unsigned rg) const { return active_[rg]; }
    size_t activeTotal() const;

private:
    std::vector<std::vector<Pair>> active_, pending_;
};

}  // namespace pdh
#endif
