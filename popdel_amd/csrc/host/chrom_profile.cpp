#include "chrom_profile.h"

#include <algorithm>
#include <atomic>
#include <stdexcept>

#include "parallel.h"
#include "profile_io.h"   // posToWin

namespace pdh {

// ----------------------------------------------------------------------------------------------------------
// StartSlots
// ----------------------------------------------------------------------------------------------------------
StartSlots::StartSlots(unsigned firstWin, unsigned windows) : numWindows(windows)
{
    slots[0].reset(0, firstWin + windows);
    slots[1].reset(0, slots[0].right + windows);
    slots[2].reset(0, slots[1].right + windows);
}

void StartSlots::switchWrite()
{
    Slot & prev = slots[writeSlot];
    Slot & next = slots[(writeSlot + 1) % 3];
    next.reset(insertionCount, prev.right + numWindows);
    writeSlot = (writeSlot + 1) % 3;
    prev.readable = true;
}

bool StartSlots::switchRead()
{
    readSlot = (readSlot + 1) % 3;
    cursorToStart();
    return !readAtEnd();
}

void StartSlots::cursorUpTo(unsigned pos)
{
    const Slot & slot = slots[readSlot];
    if (readIdx >= slot.total)
        return;
    if (!slot.isVirtual())
    {
        const HugeVector<StartEntry> & v = slot.entries;
        readIdx = static_cast<size_t>(std::upper_bound(v.begin() + static_cast<std::ptrdiff_t>(readIdx), v.end(), pos,
                                                       [](unsigned p, const StartEntry & e) { return p < e.startPos; }) - v.begin());
        return;
    }
    if (slot.backPos <= pos)
    {
        readIdx = slot.total;
        readByPos = false;
        return;
    }
    readPos = readByPos ? std::max(readPos, pos) : pos;
    readByPos = true;
}

bool StartSlots::switchBoth(unsigned & rightBorderOut)
{
    switchWrite();
    bool res = switchRead();
    rightBorderOut = slots[readSlot].right;
    return res;
}

void StartSlots::add(unsigned startPos, int deviation)
{
    slots[writeSlot].entries.push_back(StartEntry{startPos, deviation});
    slots[writeSlot].note(startPos, startPos, 1);
    ++insertionCount;
}

// Only the two slots behind the write slot can hold readable ids.
unsigned StartSlots::slotOfId(unsigned id) const
{
    unsigned newer = (writeSlot + 2) % 3;
    unsigned older = (writeSlot + 1) % 3;
    return slots[newer].idOffset <= id ? newer : older;
}

const StartEntry & StartSlots::entryOfId(unsigned id) const
{
    const Slot & s = slots[slotOfId(id)];
    return s.entries[id - s.idOffset];
}

// ----------------------------------------------------------------------------------------------------------
// EndSlots
// ----------------------------------------------------------------------------------------------------------
EndSlots::EndSlots(unsigned firstWin, unsigned windows) : numWindows(windows)
{
    slots[0].reset(firstWin + windows);
    slots[1].reset(slots[0].right + windows);
    slots[2].reset(slots[1].right + windows);
}

// The slot after the next one is recycled; the next one may already hold entries.
void EndSlots::switchWrite()
{
    Slot & cur = slots[writeSlot];
    Slot & next = slots[(writeSlot + 1) % 3];
    Slot & afterNext = slots[(writeSlot + 2) % 3];
    afterNext.reset(next.right + numWindows);
    writeSlot = (writeSlot + 1) % 3;
    cur.readable = true;
}

void EndSlots::fixConsecutiveSwitch(unsigned & activeLoad)
{
    if (endPosSlot != (writeSlot + 2) % 3)
    {
        endPosSlot = writeSlot;
        dNext = 0;
        activeLoad = 0;
    }
    else if (slots[endPosSlot].entries.empty())
    {
        endPosSlot = writeSlot;
    }
}

bool EndSlots::switchRead()
{
    readSlot = (readSlot + 1) % 3;
    slots[readSlot].readable = true;
    readIdx = 0;
    return !readAtEnd();
}

// ---- light slots ---------------------------------------------------------------------------------------------------------
namespace {
// bins of a slot's histogram: from a little below the slot's left border (right - numWindows) to well beyond its right one
// (records are filed one slot ahead when their pair ends beyond the write slot); a record outside sends the read group back
// to heavy tables
constexpr unsigned kBinsBelow = 256, kBinsAbove = 2304;
}

void EndSlots::prepareHist(Slot & slot)
{
    const unsigned nBins = numWindows / 30u + 2u + kBinsBelow + kBinsAbove;
    if (slot.hist.size() != nBins)
    {
        slot.hist.assign(nBins, 0);
        slot.lowBin = slot.highBin = 0;
        slot.peak = 0;
    }
    const unsigned leftBin = slot.right >= numWindows ? (slot.right - numWindows) / 30u : 0u;
    slot.baseBin = leftBin >= kBinsBelow ? leftBin - kBinsBelow : 0u;
}

void EndSlots::addLight(Slot & slot, unsigned id, unsigned lastWin)
{
    if (slot.total == 0 && slot.highBin == 0)
        prepareHist(slot);
    const unsigned bin = lastWin / 30u;
    if (bin < slot.baseBin || bin - slot.baseBin >= slot.hist.size() || slot.hist[bin - slot.baseBin] == 0xffffu)
    {
        // a pair that ends outside the windows a slot's counts cover (or 65,535 pairs ending in one window): this read
        // group keeps ordered tables from here on
        leaveLight();
        slot.entries.insert(std::upper_bound(slot.entries.begin(), slot.entries.end(), lastWin,
                                             [](unsigned w, const EndEntry & e) { return w < e.lastWindow; }),
                            EndEntry{id, lastWin});
        ++slot.total;
        return;
    }
    const unsigned b = bin - slot.baseBin;
    if (slot.total == 0)
    {
        slot.lowBin = b;
        slot.highBin = b + 1;
        slot.maxWin = lastWin;
    }
    else
    {
        slot.ordered = slot.ordered && lastWin >= slot.backWin;
        slot.lowBin = std::min(slot.lowBin, b);
        slot.highBin = std::max(slot.highBin, b + 1);
        slot.maxWin = std::max(slot.maxWin, lastWin);
    }
    const unsigned count = ++slot.hist[b];
    if (count > slot.peak)
        slot.peak = count;
    for (int side = 0; side < 2; ++side)
        if (b < slot.curBin[side])
            ++slot.curCum[side];
    slot.entries.push_back(EndEntry{id, lastWin});
    ++slot.total;
    slot.backWin = lastWin;
}

// ---- runs filed in bulk -----------------------------------------------------------------------------------------------------
void EndSlots::settle(Slot & slot)
{
    if (slot.deferred.empty())
        return;
    HugeVector<EndEntry> all;
    all.reserve(slot.total);
    size_t at = 0;
    for (const Deferred & d : slot.deferred)
    {
        all.insert(all.end(), slot.entries.begin() + static_cast<std::ptrdiff_t>(at), slot.entries.begin() + static_cast<std::ptrdiff_t>(d.explicitBefore));
        at = d.explicitBefore;
        if (owner && startsOf->slots[d.startSlot].isVirtual())
            owner->materializeStarts(rgIndex, d.startSlot);
        const StartEntry * src = startsOf->slots[d.startSlot].entries.data() + d.first;
        for (uint32_t j = 0; j < d.n; ++j)
        {
            const int inner = src[j].deviation + d.innerOffset;
            const unsigned endPos = src[j].startPos + (inner > 0 ? static_cast<unsigned>(inner) : 0u);
            all.push_back(EndEntry{d.firstId + j, posToWin(endPos, 30)});
        }
    }
    all.insert(all.end(), slot.entries.begin() + static_cast<std::ptrdiff_t>(at), slot.entries.end());
    slot.entries.swap(all);
    slot.deferred.clear();
}

void EndSlots::settleRefsTo(unsigned startSlot, unsigned spare)
{
    for (unsigned k = 0; k < 3; ++k)
    {
        if (k == spare)
            continue;
        for (const Deferred & d : slots[k].deferred)
            if (d.startSlot == startSlot)
            {
                settle(slots[k]);
                break;
            }
    }
}

// entries of the slot with lastWindow < pos (last windows are multiples of 30)
unsigned EndSlots::lessCount(Slot & slot, unsigned pos, int side)
{
    if (slot.total == 0)
        return 0;
    const unsigned bins = (pos + 29u) / 30u;                     // windows 0 .. bins-1 end before pos
    unsigned t = bins > slot.baseBin ? bins - slot.baseBin : 0u;
    t = std::min<unsigned>(t, static_cast<unsigned>(slot.hist.size()));
    unsigned & cb = slot.curBin[side];
    unsigned & cc = slot.curCum[side];
    if (t < cb)
    {
        cb = 0;
        cc = 0;
    }
    // (nothing below lowBin, nothing from highBin on)
    unsigned from = std::max(cb, slot.lowBin), to = std::min(t, slot.highBin);
    for (unsigned b = from; b < to; ++b)
        cc += slot.hist[b];
    cb = t;
    return cc;
}

// getEndCount() on counts: the loop over entries of the ordered array, taken a slot at a time -- entry i of the ordered
// array ends before `win` exactly if i < lessCount(win).
unsigned EndSlots::countEndedLight(unsigned pos)
{
    bool lookAhead = false;
    unsigned i = dNext;
    unsigned c = 0;
    unsigned cSlot = endPosSlot;
    unsigned n = static_cast<unsigned>(slots[cSlot].total);
    if (i == n)
    {
        if (endPosSlot != writeSlot)
        {
            endPosSlot = (endPosSlot + 1) % 3;
            dNext = dNextOffset;
            dNextOffset = 0;
            i = dNext;
            cSlot = endPosSlot;
            n = static_cast<unsigned>(slots[cSlot].total);
        }
        else
        {
            return 0;
        }
    }
    if (n == 0)
        return 0;
    const unsigned win = posToWin(pos);
    for (;;)
    {
        const unsigned below = lessCount(slots[cSlot], win, 0);
        if (!(i < n && i < below))
            break;
        const unsigned k = std::min(n, below) - i;
        if (lookAhead)
            dNextOffset += k;
        c += k;
        i += k;
        if (i != n)
            break;
        if (endPosSlot != writeSlot)
        {
            endPosSlot = (endPosSlot + 1) % 3;
            dNext = dNextOffset;
            dNextOffset = 0;
            return c;
        }
        cSlot = (cSlot + 1) % 3;
        n = static_cast<unsigned>(slots[cSlot].total);
        if (n == 0)
            break;
        i = dNextOffset;
        lookAhead = true;
    }
    dNext += c - dNextOffset;
    return c;
}

void EndSlots::leaveLight()
{
    if (!light)
        return;
    light = false;
    for (Slot & s : slots)
        settle(s);                      // (runs filed in bulk: their entries are needed from here on)
    for (Slot & s : slots)
        if (!s.ordered)
        {
            std::stable_sort(s.entries.begin(), s.entries.end(), [](const EndEntry & a, const EndEntry & b) { return a.lastWindow < b.lastWindow; });
            s.ordered = true;
        }
}

void EndSlots::setLight(bool on)
{
    if (on == light)
        return;
    if (!on)
    {
        leaveLight();
        return;
    }
    light = true;
    for (Slot & s : slots)
    {
        // (entries are ordered: heavy tables) -> counts
        for (unsigned b = s.lowBin; b < s.highBin && b < s.hist.size(); ++b)
            s.hist[b] = 0;
        s.lowBin = s.highBin = 0;
        s.peak = 0;
        s.curBin[0] = s.curBin[1] = s.curCum[0] = s.curCum[1] = 0;
        s.ordered = true;
        if (s.entries.empty())
            continue;
        prepareHist(s);
        HugeVector<EndEntry> kept;
        kept.swap(s.entries);
        s.total = 0;
        for (const EndEntry & e : kept)
        {
            addLight(s, e.id, e.lastWindow);
            if (!light)                         // (a record outside the counts' range: this read group stays heavy)
            {
                s.entries.swap(kept);
                s.total = s.entries.size();
                return;
            }
        }
    }
}

void EndSlots::advanceReadTo(unsigned pos, unsigned shift)
{
    Slot & slot = slots[readSlot];
    const size_t n = slot.total;
    const size_t from = std::min(readIdx, n);
    if (light)
    {
        const size_t over = pos > shift ? lessCount(slot, pos - shift, 1) : 0u;
        readIdx = std::max(from, over);
        return;
    }
    readIdx = static_cast<size_t>(std::lower_bound(slot.entries.begin() + static_cast<std::ptrdiff_t>(from), slot.entries.end(), pos,
                                                   [shift](const EndEntry & e, unsigned p) { return e.lastWindow + shift < p; }) -
                                  slot.entries.begin());
}

unsigned EndSlots::lastWindowOfReadSlot() const
{
    const Slot & slot = slots[readSlot];
    return light ? slot.maxWin : slot.entries.back().lastWindow;
}

void EndSlots::add(unsigned id, unsigned endPos)
{
    const unsigned lastWin = posToWin(endPos, 30);
    unsigned which = writeSlot;
    if (needsSwitch(lastWin))
        which = (which + 1) % 3;
    if (light)
    {
        addLight(slots[which], id, lastWin);
        return;
    }
    // behind the last entry that does not end later (what upper_bound finds) -- looked for from the back: pairs arrive in
    // start order, so a new pair's last window lies among the last few entries' (a binary search over the whole slot
    // costs more than the handful of comparisons this takes)
    HugeVector<EndEntry> & v = slots[which].entries;
    size_t at = v.size();
    while (at > 0 && v[at - 1].lastWindow > lastWin)
        --at;
    v.insert(v.begin() + static_cast<std::ptrdiff_t>(at), EndEntry{id, lastWin});
    ++slots[which].total;
}

// Number of pairs of the current end-position slot (and, by look-ahead, of the following one) whose last
// window lies before the window of `pos`; advances the dNext cursor. Reference: getEndCount(), :870-924.
unsigned EndSlots::countEnded(unsigned pos)
{
    if (light)
        return countEndedLight(pos);
    bool lookAhead = false;
    unsigned i = dNext;
    unsigned c = 0;
    unsigned cSlot = endPosSlot;
    unsigned n = static_cast<unsigned>(slots[cSlot].entries.size());
    if (i == n)
    {
        if (endPosSlot != writeSlot)
        {
            endPosSlot = (endPosSlot + 1) % 3;
            dNext = dNextOffset;
            dNextOffset = 0;
            i = dNext;
            cSlot = endPosSlot;
            n = static_cast<unsigned>(slots[cSlot].entries.size());
        }
        else
        {
            return 0;
        }
    }
    if (n == 0)
        return 0;
    const unsigned win = posToWin(pos);
    while (i < n && slots[cSlot].entries[i].lastWindow < win)
    {
        if (lookAhead)
            ++dNextOffset;
        ++c;
        ++i;
        if (i == n)
        {
            if (endPosSlot != writeSlot)
            {
                endPosSlot = (endPosSlot + 1) % 3;
                dNext = dNextOffset;
                dNextOffset = 0;
                return c;
            }
            cSlot = (cSlot + 1) % 3;
            n = static_cast<unsigned>(slots[cSlot].entries.size());
            if (n == 0)
                break;
            i = dNextOffset;
            lookAhead = true;
        }
    }
    dNext += c - dNextOffset;
    return c;
}

// ----------------------------------------------------------------------------------------------------------
// ChromProfile
// ----------------------------------------------------------------------------------------------------------
ChromProfile::ChromProfile(unsigned numReadGroups, const std::vector<unsigned> & maximumLoad, unsigned bufferSize)
    : starts(numReadGroups, StartSlots(0, bufferSize)),
      ends(numReadGroups, EndSlots(0, bufferSize)),
      active(numReadGroups),
      activeLoad(numReadGroups, 0),
      maxLoad(maximumLoad),
      switches_(numReadGroups, 2)
{
    currentRightBorder = starts[0].slots[2].right;
    log_.resize(numReadGroups);
    for (unsigned rg = 0; rg < numReadGroups; ++rg)
    {
        ends[rg].startsOf = &starts[rg];            // (neither vector is ever resized)
        ends[rg].owner = this;
        ends[rg].rgIndex = rg;
    }
}

// What the closed-form walk needs to know about a start slot beyond its entries: under which end slot's border their end
// records are filed (constant while the slot is written), and that the entries arrive in start order.
static inline void noteAdd(StartSlots & st, const EndSlots & en, unsigned startPos)
{
    StartSlots::Slot & slot = st.slots[st.writeSlot];
    slot.endBorder = en.slots[en.writeSlot].right;
    (void)startPos;                     // (whether the entries arrive in order: Slot::note, with the entry itself)
}

void ChromProfile::add(unsigned rg, unsigned startPos, unsigned endPos, int deviation)
{
    StartSlots & st = starts[rg];
    EndSlots & en = ends[rg];
    const unsigned id = st.insertionCount;
    if (activeLoad[rg] >= maxLoad[rg])
    {
        // Saturated: only look for closing pairs, add the new pair if that freed room.
        unsigned closing = en.countEnded(startPos);
        activeLoad[rg] -= closing;
        if (activeLoad[rg] < maxLoad[rg])
        {
            noteAdd(st, en, startPos);
            st.add(startPos, deviation);
            en.add(id, endPos);
            ++activeLoad[rg];
        }
    }
    else
    {
        noteAdd(st, en, startPos);
        st.add(startPos, deviation);
        en.add(id, endPos);
        ++activeLoad[rg];
        unsigned closing = en.countEnded(startPos);
        activeLoad[rg] -= closing;
    }
}

void ChromProfile::reserveRun(unsigned rg, size_t n)
{
    // (with room to spare: a slot that already holds a round's entries is copied when it has to grow, and record counts
    // drift by a few per cent from round to round -- an exact reserve moved 170 KB per table and sample in most rounds)
    HugeVector<StartEntry> & s = starts[rg].slots[starts[rg].writeSlot].entries;
    if (s.capacity() < s.size() + n)
        s.reserve(std::max(s.size() + n + n / 4, s.capacity() + s.capacity() / 2));
    HugeVector<EndEntry> & e = ends[rg].slots[ends[rg].writeSlot].entries;
    if (e.capacity() < e.size() + n)
        e.reserve(std::max(e.size() + n + n / 4, e.capacity() + e.capacity() / 2));
}

void ChromProfile::reserveStarts(unsigned rg, size_t n)
{
    HugeVector<StartEntry> & s = starts[rg].slots[starts[rg].writeSlot].entries;
    if (s.capacity() < s.size() + n)
        s.reserve(std::max(s.size() + n + n / 4, s.capacity() + s.capacity() / 2));
}

void ChromProfile::addRun(unsigned rg, const uint32_t * posDev, size_t n, int innerOffset)
{
    StartSlots & st = starts[rg];
    EndSlots & en = ends[rg];
    // (the load counter is kept in a register over the run: the counters of neighbouring read groups share cache lines,
    // and neighbouring samples are replayed by different threads -- updating it in place per record cost half of the
    // replay phase in false sharing)
    unsigned load = activeLoad[rg];
    const unsigned cap = maxLoad[rg];
    if (en.light)
    {
        // the same operations with the everyday case of each inlined (EndSlots::addFast / countEndedFast)
        auto file = [&](unsigned startPos, int deviation, unsigned endPos) {
            StartSlots::Slot & slot = st.slots[st.writeSlot];
            slot.endBorder = en.slots[en.writeSlot].right;
            slot.note(startPos, startPos, 1);
            const unsigned id = st.insertionCount++;
            slot.entries.push_back(StartEntry{startPos, deviation});
            en.addFast(id, endPos);
        };
        auto one = [&](size_t i) {
            const unsigned startPos = posDev[2 * i];
            const int deviation = static_cast<int32_t>(posDev[2 * i + 1]);
            const int inner = deviation + innerOffset;
            const unsigned endPos = startPos + (inner > 0 ? static_cast<unsigned>(inner) : 0u);
            if (load >= cap)
            {
                load -= en.countEndedFast(startPos);
                if (load < cap)
                {
                    file(startPos, deviation, endPos);
                    ++load;
                }
            }
            else
            {
                file(startPos, deviation, endPos);
                ++load;
                load -= en.countEndedFast(startPos);
            }
        };
        // Record by record the rule asks the end table after every record how many pairs are over. Where that cannot matter --
        // a block of records short enough for the load to stay below the cap whatever the answers, in non-decreasing order,
        // with the counting cursor certain to stay inside its slot (EndSlots::quietBlock) -- the block is filed in one go and the
        // table asked once, behind its last record: the answers add up to the same number and leave the same cursors.
        constexpr size_t kBlock = 256;
        unsigned endBuf[kBlock];        // the records' last windows, as window numbers (end position / 30)
        size_t i = 0;
        while (i < n)
        {
            size_t block = 0;
            bool below = true;
            if (load < cap && cap - load >= 8u && n - i >= 8u)
            {
                // the block: as many records as the cap leaves room for, as long as they come in order (and, if the end table's
                // counting cursor is in its write slot, as long as they end below that slot's border)
                const size_t want = std::min<size_t>(std::min<size_t>(kBlock, cap - load), n - i);
                const unsigned borderBin = (en.slots[en.writeSlot].right + 29u) / 30u;      // (a last window 30 * bin lies below the border iff bin < borderBin)
                const bool cursorInWriteSlot = en.endPosSlot == en.writeSlot;
                unsigned prev = posDev[2 * i];
                for (; block < want; ++block)
                {
                    const unsigned startPos = posDev[2 * (i + block)];
                    const int inner = static_cast<int32_t>(posDev[2 * (i + block) + 1]) + innerOffset;
                    const unsigned bin = (startPos + (inner > 0 ? static_cast<unsigned>(inner) : 0u)) / 30u;
                    const bool in = bin < borderBin;
                    if (startPos < prev || (cursorInWriteSlot && !in && block >= 4))
                        break;
                    below = below && in;
                    prev = startPos;
                    endBuf[block] = bin;
                }
                if (block >= 4)
                    block = en.quietBlock(posDev[2 * i], posDev[2 * (i + block - 1)], block, below);
            }
            if (block < 4)
            {
                // (not now: a few records one by one before the next try)
                const size_t upTo = std::min(n, i + 8);
                for (; i < upTo; ++i)
                    one(i);
                continue;
            }
            StartSlots::Slot & slot = st.slots[st.writeSlot];
            slot.endBorder = en.slots[en.writeSlot].right;
            slot.note(posDev[2 * i], posDev[2 * (i + block - 1)], block);
            const unsigned id = st.insertionCount;
            st.insertionCount += static_cast<unsigned>(block);
            static_assert(sizeof(StartEntry) == 2 * sizeof(uint32_t), "a record {pos, dev} is a start entry");
            const StartEntry * src = reinterpret_cast<const StartEntry *>(posDev + 2 * i);
            slot.entries.insert(slot.entries.end(), src, src + block);
            if (below)
                en.addBlockToWriteSlot(id, endBuf, block);
            else
                for (size_t j = 0; j < block; ++j)
                    en.addFast(id + static_cast<unsigned>(j), endBuf[j] * 30u);     // (any end position inside the window files the same)
            load += static_cast<unsigned>(block);
            load -= en.countEndedFast(posDev[2 * (i + block - 1)]);
            i += block;
        }
        activeLoad[rg] = load;
        return;
    }
    for (size_t i = 0; i < n; ++i)
    {
        const unsigned startPos = posDev[2 * i];
        const int deviation = static_cast<int32_t>(posDev[2 * i + 1]);
        const int inner = deviation + innerOffset;
        const unsigned endPos = startPos + (inner > 0 ? static_cast<unsigned>(inner) : 0u);
        const unsigned id = st.insertionCount;
        if (load >= cap)
        {
            load -= en.countEnded(startPos);
            if (load < cap)
            {
                noteAdd(st, en, startPos);
                st.add(startPos, deviation);
                en.add(id, endPos);
                ++load;
            }
        }
        else
        {
            noteAdd(st, en, startPos);
            st.add(startPos, deviation);
            en.add(id, endPos);
            ++load;
            load -= en.countEnded(startPos);
        }
    }
    activeLoad[rg] = load;
}

void summarizeRun(const uint32_t * posDev, size_t nRecords, int innerOffset, uint32_t borderBin, uint32_t roomStarts, uint32_t roomEnds,
                  RunSummary & out, uint8_t * starts, uint8_t * ends)
{
    out = RunSummary();
    if (nRecords == 0)
        return;
    auto endBinOf = [&](size_t i) {
        const int inner = static_cast<int32_t>(posDev[2 * i + 1]) + innerOffset;
        return (posDev[2 * i] + (inner > 0 ? static_cast<unsigned>(inner) : 0u)) / 30u;
    };
    size_t n = 0;
    while (n < nRecords && endBinOf(n) < borderBin)
        ++n;
    out.n = static_cast<uint32_t>(n);
    if (n == 0)
        return;
    out.firstPos = posDev[0];
    out.lastPos = posDev[2 * (n - 1)];
    out.binBase = out.firstPos / 30u;
    out.firstEndBin = endBinOf(0);
    out.lastEndBin = endBinOf(n - 1);
    out.minEndBin = 0xffffffffu;
    out.maxEndBin = 0;
    out.flags = RunSummary::kStartsInOrder | RunSummary::kEndsInOrder;
    std::fill(starts, starts + roomStarts, static_cast<uint8_t>(0));
    std::fill(ends, ends + roomEnds, static_cast<uint8_t>(0));
    uint32_t prevPos = out.firstPos, prevEnd = out.firstEndBin;
    for (size_t i = 0; i < n; ++i)
    {
        const uint32_t pos = posDev[2 * i], sb = pos / 30u, eb = endBinOf(i);
        if (pos < prevPos)
            out.flags &= ~RunSummary::kStartsInOrder;
        if (eb < prevEnd)
            out.flags &= ~RunSummary::kEndsInOrder;
        prevPos = pos;
        prevEnd = eb;
        out.minEndBin = std::min(out.minEndBin, eb);
        out.maxEndBin = std::max(out.maxEndBin, eb);
        // (a start below the first one's window, or windows beyond the room: flagged, the counts are not used then)
        if (sb < out.binBase || sb - out.binBase >= roomStarts || eb < out.binBase || eb - out.binBase >= roomEnds)
        {
            out.flags |= RunSummary::kNoRoom;
            continue;
        }
        uint8_t & cs = starts[sb - out.binBase];
        uint8_t & ce = ends[eb - out.binBase];
        if (cs == 255u || ce == 255u)
            out.flags |= RunSummary::kSaturated;
        if (cs != 255u)
            ++cs;
        if (ce != 255u)
            ++ce;
    }
    // (a count that reached 255 may have missed records: saturated either way)
    out.nBins = (out.flags & RunSummary::kNoRoom) ? 0u : out.maxEndBin - out.binBase + 1u;
    for (uint32_t b = 0; b < roomStarts; ++b)
        if (starts[b] == 255u)
            out.flags |= RunSummary::kSaturated;
    for (uint32_t b = 0; b < roomEnds; ++b)
        if (ends[b] == 255u)
            out.flags |= RunSummary::kSaturated;
}

bool ChromProfile::addBulk(unsigned rg, const uint32_t * posDev, const RunSummary & s, const uint16_t * startCounts, const uint16_t * endCounts,
                           int innerOffset, const VirtualNote & note)
{
    StartSlots & st = starts[rg];
    EndSlots & en = ends[rg];
    const unsigned cap = maxLoad[rg];
    const unsigned load0 = activeLoad[rg];
    constexpr uint32_t kWanted = RunSummary::kStartsInOrder;
    if (!en.light || s.n == 0 || (s.flags & kWanted) != kWanted || (s.flags & (RunSummary::kSaturated | RunSummary::kNoRoom)) != 0 || load0 >= cap)
        return false;
    // ---- the end table: the counting cursor at home in the write slot, in the everyday state, and it stays there
    if (en.endPosSlot != en.writeSlot || en.dNextOffset != 0)
        return false;
    EndSlots::Slot & slot = en.slots[en.writeSlot];
    const size_t have = slot.total;
    if (en.dNext >= have)                       // (also: the slot has entries -- its counts are set up)
        return false;
    const unsigned base = slot.baseBin;
    const unsigned firstBin = s.firstPos / 30u, lastBin = s.lastPos / 30u;
    if (firstBin != s.binBase || lastBin < firstBin || lastBin - s.binBase >= s.nBins)
        return false;
    const unsigned tFirst = firstBin > base ? firstBin - base : 0u;
    if (tFirst < slot.curBin[0] || lastBin < base || lastBin - base > slot.hist.size())
        return false;
    // every pair ends below the write slot's border (the summary was cut there) and inside the windows the counts cover
    const unsigned borderBin = (slot.right + 29u) / 30u;
    if (s.maxEndBin >= borderBin || s.minEndBin < base || s.maxEndBin - base >= slot.hist.size() || s.minEndBin < s.binBase ||
        s.maxEndBin - s.binBase >= s.nBins)
        return false;
    if (slot.peak >= 0x8000u)                   // (no count of the slot may saturate: every count of the run is below 0x7fff, checked below)
        return false;
    // ---- the load behind every record, window by window: behind the last record that begins in window b the counting cursor
    // stands at max(where it stood, entries of the slot that end before b) -- the records of the run that end before b began
    // before b, so they are filed by then -- and the load is what it was, plus the records filed, minus the cursor's way
    uint64_t belowAtEnd = 0;                    // entries that end before the window of the run's last record, once it is filed
    {
        const uint16_t * hist = slot.hist.data();
        uint64_t oldBelow = slot.curCum[0];     // entries already in the slot that end before window (base + at)
        unsigned at = slot.curBin[0];
        const uint64_t dNext0 = en.dNext;
        const unsigned histBins = static_cast<unsigned>(slot.hist.size());
        // (the prefix over the slot's own counts up to the run's first window: the slot has entries in [lowBin, highBin) only)
        {
            const unsigned upTo = firstBin > base ? std::min(firstBin - base, histBins) : 0u;
            for (const unsigned stop = std::min(upTo, slot.highBin); at < stop; ++at)
                if (at >= slot.lowBin)
                    oldBelow += hist[at];
            at = std::max(at, upTo);
        }
        // From here on a window per trip, nothing to decide inside: window b's records are filed behind `filed` records of the
        // run, `below` entries end before it -- the slot's (the prefix goes on by hist[t - 1], taken as zero outside the slot's
        // range by cutting the loop there) and the run's --, the load behind them is load0 + filed - (max(below, dNext0) - dNext0).
        // A window without records repeats the question of the window before it with a cursor further on: never a larger load.
        // The largest load decides at the end (the rule bites rarely: nothing is gained by leaving early).
        uint64_t runBelow = 0, filed = 0, worst = 0;
        unsigned tooMany = 0;
        const uint16_t * sc = startCounts + (firstBin - s.binBase), * ec = endCounts + (firstBin - s.binBase);
        const unsigned nWin = lastBin - firstBin + 1u;
        // windows whose predecessor lies inside the slot's counts: t - 1 = (firstBin + w) - base - 1 in [max(at', lowBin), highBin)
        // with at' = where the prefix stands; w from wLo to wHi
        const int64_t tFirstSigned = static_cast<int64_t>(firstBin) - static_cast<int64_t>(base);
        auto clampW = [&](int64_t t) { return static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>(t - tFirstSigned, 0), nWin)); };
        // the prefix `oldBelow` holds hist[0 .. at); window w needs hist[0 .. min(t_w, histBins)) with t_w = tFirstSigned + w
        const unsigned wLo = clampW(static_cast<int64_t>(std::max(at, slot.lowBin)) + 1), wHi = std::max(wLo, clampW(static_cast<int64_t>(std::min(slot.highBin, histBins)) + 1));
        auto sweep = [&](unsigned w0, unsigned w1, bool inside) {
            // (own[w] = hist[t_w - 1]; the pointer is formed only where such windows exist: w0 + tFirstSigned >= 1 there)
            const uint16_t * own = inside && w0 < w1 ? hist + (tFirstSigned - 1 + static_cast<int64_t>(w0)) - w0 : nullptr;
            for (unsigned w = w0; w < w1; ++w)
            {
                if (own)
                    oldBelow += own[w];
                filed += sc[w];
                const uint64_t below = oldBelow + runBelow;
                const uint64_t load = load0 + filed - (std::max(below, dNext0) - dNext0);
                worst = std::max(worst, load);
                tooMany |= ec[w];
                runBelow += ec[w];
            }
        };
        sweep(0, wLo, false);
        sweep(wLo, wHi, true);
        sweep(wHi, nWin, false);
        belowAtEnd = oldBelow + runBelow - ec[nWin - 1u];
        if (worst >= cap)
            return false;
        for (unsigned b = lastBin + 1; b <= s.maxEndBin; ++b)
            tooMany |= endCounts[b - s.binBase];
        if (filed != s.n || tooMany >= 0x4000u)
            return false;                       // (the counts do not add up to the records, or one of them is beyond any sum of bytes: not a summary of this run)
    }
    // ---- file: the start entries in one copy, the end entries as counts and a reference
    StartSlots::Slot & sslot = st.slots[st.writeSlot];
    sslot.endBorder = slot.right;
    const unsigned id = st.insertionCount;
    st.insertionCount += s.n;
    const uint32_t firstEntry = static_cast<uint32_t>(sslot.total);         // (its place among ALL the slot's entries)
    sslot.note(s.firstPos, s.lastPos, s.n);
    if (posDev)
    {
        const StartEntry * src = reinterpret_cast<const StartEntry *>(posDev);
        sslot.entries.insert(sslot.entries.end(), src, src + s.n);
    }
    else
        sslot.virtualRuns.push_back(StartSlots::VirtualRun{sslot.entries.size(), s.n, note.firstWindow, note.skip, note.jumpPos, note.tag});
    {
        // (three plain loops: the counts added, and what of them lies in front of the two cursors)
        const unsigned cur0 = slot.curBin[0], cur1 = slot.curBin[1];
        const unsigned t0 = s.minEndBin - base, nEnd = s.maxEndBin - s.minEndBin + 1u;
        uint16_t * __restrict__ own = slot.hist.data() + t0;
        const uint16_t * __restrict__ add = endCounts + (s.minEndBin - s.binBase);
        unsigned cum0 = 0, cum1 = 0;
        uint16_t peak16 = 0;
        for (unsigned k = 0; k < nEnd; ++k)
        {
            const uint16_t count = static_cast<uint16_t>(own[k] + add[k]);
            own[k] = count;
            peak16 = count > peak16 ? count : peak16;
        }
        const unsigned peak = std::max<unsigned>(slot.peak, peak16);
        for (unsigned k = 0, n0 = cur0 > t0 ? std::min(cur0 - t0, nEnd) : 0u; k < n0; ++k)
            cum0 += add[k];
        for (unsigned k = 0, n1 = cur1 > t0 ? std::min(cur1 - t0, nEnd) : 0u; k < n1; ++k)
            cum1 += add[k];
        slot.peak = peak;
        slot.curCum[0] += cum0;
        slot.curCum[1] += cum1;
        slot.ordered = slot.ordered && (s.flags & RunSummary::kEndsInOrder) != 0 && s.firstEndBin * 30u >= slot.backWin;
        slot.backWin = s.lastEndBin * 30u;
        if (s.minEndBin - base < slot.lowBin)
            slot.lowBin = s.minEndBin - base;
        if (s.maxEndBin - base >= slot.highBin)
            slot.highBin = s.maxEndBin - base + 1;
        if (s.maxEndBin * 30u > slot.maxWin)
            slot.maxWin = s.maxEndBin * 30u;
        slot.deferred.push_back(EndSlots::Deferred{slot.entries.size(), st.writeSlot, firstEntry, s.n, id, innerOffset});
        slot.total += s.n;
    }
    // ... and the question behind the run's last record (EndSlots::countEndedFast(s.lastPos), whose everyday case this is by the
    // checks above): the cursor over the counts goes to that record's window, where belowAtEnd entries lie before it
    unsigned load = load0 + s.n;
    {
        const unsigned tLast = lastBin > base ? lastBin - base : 0u;
        slot.curBin[0] = tLast;
        slot.curCum[0] = static_cast<unsigned>(belowAtEnd);
        if (belowAtEnd > en.dNext)
        {
            const unsigned k = static_cast<unsigned>(belowAtEnd - en.dNext);
            en.dNext += k;                      // (below the slot's entries: the record filed last is not over)
            load -= k;
        }
    }
    activeLoad[rg] = load;
    return true;
}

void ChromProfile::materializeStarts(unsigned rg, unsigned slotIdx)
{
    StartSlots & st = starts[rg];
    StartSlots::Slot & slot = st.slots[slotIdx];
    if (!slot.isVirtual())
        return;
    if (!refill)
        throw std::runtime_error("[PopDel] Internal error: a start slot holds runs of entries that nobody can read back.");
    HugeVector<StartEntry> all;
    all.resize(slot.total);
    size_t at = 0, out = 0;
    for (const StartSlots::VirtualRun & run : slot.virtualRuns)
    {
        std::copy(slot.entries.begin() + static_cast<std::ptrdiff_t>(at), slot.entries.begin() + static_cast<std::ptrdiff_t>(run.explicitBefore),
                  all.begin() + static_cast<std::ptrdiff_t>(out));
        out += run.explicitBefore - at;
        at = run.explicitBefore;
        refill(rg, run, all.data() + out);
        out += run.n;
        materialized_ += run.n;
    }
    std::copy(slot.entries.begin() + static_cast<std::ptrdiff_t>(at), slot.entries.end(), all.begin() + static_cast<std::ptrdiff_t>(out));
    slot.entries.swap(all);
    slot.virtualRuns.clear();
    if (slotIdx == st.readSlot && st.readByPos)
    {
        // the cursor as a number again: behind every entry that starts at or before readPos (and not before where it stood)
        const HugeVector<StartEntry> & v = slot.entries;
        const size_t from = std::min(st.readIdx, v.size());
        st.readIdx = static_cast<size_t>(std::upper_bound(v.begin() + static_cast<std::ptrdiff_t>(from), v.end(), st.readPos,
                                                          [](unsigned p, const StartEntry & e) { return p < e.startPos; }) - v.begin());
        st.readByPos = false;
    }
}

void ChromProfile::materializeAllStarts()
{
    parallelFor(starts.size(), [&](size_t rg) {
        for (unsigned k = 0; k < 3; ++k)
            materializeStarts(static_cast<unsigned>(rg), k);
    }, 16);
}

void ChromProfile::activate(unsigned rg, unsigned id, unsigned logWindow)
{
    auto & set = active[rg];
    if (set.find(id) != nullptr)
        return;
    const StartEntry & e = starts[rg].entryOfId(id);
    ActiveRec rec{e.deviation, e.startPos, kNoLog};
    if (logging_)
    {
        rec.logIdx = static_cast<uint32_t>(log_[rg].size());
        log_[rg].push_back(PairInstance{e.deviation, e.startPos, logWindow, kNoLog, id});
    }
    if (opLogging_)
        ops_[rg].push_back(ActiveOp{logging_ ? logWindow : 0u, id, 1u});
    set.emplace(id, rec);
}

void ChromProfile::deactivate(unsigned rg, unsigned id, unsigned logWindow)
{
    auto & set = active[rg];
    const ActiveRec * rec = set.find(id);
    if (rec == nullptr)
        return;
    if (logging_ && rec->logIdx != kNoLog)
        log_[rg][rec->logIdx].off = logWindow;
    if (opLogging_)
        ops_[rg].push_back(ActiveOp{logging_ ? logWindow : 0u, id, 0u});
    set.erase(id);
}

void ChromProfile::initializeActive()
{
    for (unsigned rg = 0; rg < starts.size(); ++rg)
    {
        StartSlots & st = starts[rg];
        st.readIdx = 0;
        if (st.readAtEnd())
            continue;
        unsigned pos = st.current().startPos;
        if (pos < currentPos)
            currentPos = pos;
    }
    globalMinPos = currentPos;
    for (unsigned rg = 0; rg < starts.size(); ++rg)
    {
        StartSlots & st = starts[rg];
        if (st.readAtEnd() || !active[rg].empty())
            continue;
        const unsigned pos = st.current().startPos;
        if (pos != currentPos)
            continue;
        unsigned id = st.slots[st.readSlot].idOffset;
        activate(rg, id, logWindow_);
        ++id;
        bool endOfSlot = !st.goNext();
        while (!endOfSlot && st.current().startPos == pos)
        {
            activate(rg, id, logWindow_);
            ++id;
            endOfSlot = !st.goNext();
        }
    }
}

void ChromProfile::nextStartWindow(unsigned rg, unsigned logWindow)
{
    StartSlots & st = starts[rg];
    const unsigned pos = st.current().startPos;
    bool more = true;
    while (more && pos == st.current().startPos)
    {
        activate(rg, st.currentId(), logWindow);
        more = st.goNext();
    }
}

bool ChromProfile::nextEndWindow(unsigned rg, unsigned logWindow)
{
    EndSlots & en = ends[rg];
    if (en.readAtEnd())
        return false;
    const unsigned win = en.current().lastWindow;
    while (win == en.current().lastWindow)
    {
        deactivate(rg, en.current().id, logWindow);
        if (!en.goNext())
            return false;
    }
    return true;
}

bool ChromProfile::nextWindow(unsigned shift)
{
    bool someStillGood = false;
    currentPos += shift;
    if (currentPos >= currentRightBorder)
    {
        currentPos -= shift;
        return false;
    }
    for (unsigned rg = 0; rg < starts.size(); ++rg)
    {
        StartSlots & st = starts[rg];
        EndSlots & en = ends[rg];
        if (!(en.readAtEnd() && st.readAtEnd()))
            someStillGood = true;
        while (!st.readAtEnd())
        {
            if (st.current().startPos <= currentPos)
                nextStartWindow(rg, logWindow_);
            else
                break;
        }
        while (!en.readAtEnd())
        {
            if (en.current().lastWindow + shift < currentPos)   // the last window itself is still valid
                nextEndWindow(rg, logWindow_);
            else
                break;
        }
    }
    if (!someStillGood)
        currentPos -= shift;
    return someStillGood;
}

// One read group's share of the window loop: steps 1..maxSteps at positions pos0 + j * shift, until both of its
// cursors are at the end. Returns the number of steps at whose start the read group still had entries.
unsigned ChromProfile::walkReadGroup(unsigned rg, unsigned pos0, unsigned shift, unsigned maxSteps)
{
    StartSlots & st = starts[rg];
    EndSlots & en = ends[rg];
    unsigned j = 0;
    while (j < maxSteps && !(en.readAtEnd() && st.readAtEnd()))
    {
        ++j;
        const unsigned pos = pos0 + j * shift;
        while (!st.readAtEnd())
        {
            if (st.current().startPos <= pos)
                nextStartWindow(rg, j);
            else
                break;
        }
        while (!en.readAtEnd())
        {
            if (en.current().lastWindow + shift < pos)
                nextEndWindow(rg, j);
            else
                break;
        }
    }
    return j;
}

unsigned ChromProfile::walkSegment(unsigned shift)
{
    if (profilesAtEnd)
        return logWindow_;
    const unsigned pos0 = currentPos;
    // steps the slot border allows: positions pos0 + j * shift must stay below currentRightBorder
    unsigned maxSteps = 0;
    if (pos0 + shift < currentRightBorder)
        maxSteps = (currentRightBorder - pos0 - 1) / shift;
    std::vector<unsigned> good(starts.size(), 0);
    // (pointer chasing through the slot tables and the active sets of one read group at a time: on the 256-core GPU host
    // 16 / 32 / 64 threads measured 0.12 / 0.13 / 0.30 s for 1000 samples x 1 Mb)
    parallelFor(starts.size(), [&](size_t rg) { good[rg] = walkReadGroup(static_cast<unsigned>(rg), pos0, shift, maxSteps); }, 8, 24);
    unsigned steps = 0;
    for (unsigned g : good)
        steps = std::max(steps, g);
    currentPos = pos0 + steps * shift;
    profilesAtEnd = true;
    logWindow_ += steps + 1;
    return logWindow_;
}

void ChromProfile::resetTo(unsigned pos)
{
    const unsigned w = starts[0].numWindows;
    const unsigned r1 = pos + w, r2 = pos + 2 * w, r3 = pos + 3 * w;
    for (StartSlots & st : starts)
    {
        unsigned idO = st.insertionCount;
        st.slots[0].reset(idO, r1);
        st.slots[1].reset(idO, r2);
        st.slots[2].reset(idO, r3);
        st.writeSlot = 0;
        st.readSlot = 2;
        st.cursorToStart();
    }
    for (EndSlots & en : ends)
    {
        en.slots[0].reset(r1);
        en.slots[1].reset(r2);
        en.slots[2].reset(r3);
        en.writeSlot = 0;
        en.readSlot = 2;
        en.readIdx = 0;
        en.endPosSlot = 0;
        en.dNext = 0;
        en.dNextOffset = 0;
    }
    currentRightBorder = starts[0].slots[2].right;
    for (unsigned & l : activeLoad)
        l = 0;
    for (uint8_t & c : switches_)
        c = 2;                              // the tables are gone: whoever evaluates the walk drops pending and filed-ahead pairs
}

void ChromProfile::fullReset()
{
    for (StartSlots & st : starts)
        st.insertionCount = 0;
    resetTo(0);
    globalMinPos = std::numeric_limits<unsigned>::max();
    currentPos = std::numeric_limits<unsigned>::max();
    profilesAtEnd = false;
    currentRightBorder = starts[0].slots[2].right;
}

// A slot switch drains the end read slot: every pair whose end record sits there leaves the active set. With light tables
// nobody is in the active sets (the window loop runs elsewhere): the cursor simply goes to the end.
void ChromProfile::drainEndReadSlot(unsigned rg)
{
    EndSlots & en = ends[rg];
    if (en.light)
    {
        en.readIdx = en.slots[en.readSlot].total;
        return;
    }
    while (!en.readAtEnd())
        nextEndWindow(rg, logWindow_);
}

void ChromProfile::setLightEnds(bool on)
{
    lightEnds_ = on;
    parallelFor(ends.size(), [&](size_t rg) { ends[rg].setLight(on); }, 64);
}

void ChromProfile::performSwitches(const std::vector<unsigned> & rgs, unsigned * border)
{
    unsigned & rightBorder = border ? *border : currentRightBorder;
    for (unsigned i : rgs)
    {
        drainEndReadSlot(i);
        // (the start slot that is recycled may still back end entries filed in bulk: written out first -- except those of
        // the end slot that is reset in the same breath)
        if (ends[i].hasDeferred())
            ends[i].settleRefsTo((starts[i].writeSlot + 1) % 3, (ends[i].writeSlot + 2) % 3);
        starts[i].switchBoth(rightBorder);
        ends[i].switchWrite();
        ends[i].fixConsecutiveSwitch(activeLoad[i]);
        ends[i].switchRead();
        switches_[i] = static_cast<uint8_t>(std::min(2, switches_[i] + 1));
    }
}

void ChromProfile::performPartialSwitches(const std::vector<unsigned> & rgs, unsigned * border)
{
    unsigned & rightBorder = border ? *border : currentRightBorder;
    for (unsigned i : rgs)
    {
        drainEndReadSlot(i);
        if (ends[i].hasDeferred())
            ends[i].settleRefsTo((starts[i].writeSlot + 1) % 3, 3u);
        starts[i].switchBoth(rightBorder);
        ends[i].switchRead();
        switches_[i] = static_cast<uint8_t>(std::min(2, switches_[i] + 1));
    }
}

bool ChromProfile::allEmpty() const
{
    for (size_t i = 0; i < starts.size(); ++i)
    {
        if (starts[i].slots[starts[i].writeSlot].total != 0)
            return false;
        if (ends[i].slots[ends[i].writeSlot].total != 0)
            return false;
        if (!active[i].empty())
            return false;
    }
    if (externalActive && externalActive() != 0)
        return false;
    return true;
}

void ChromProfile::beginSegmentLog()
{
    logging_ = true;
    logWindow_ = 0;
    std::vector<std::pair<uint32_t, ActiveRec *>> carried;
    for (size_t rg = 0; rg < active.size(); ++rg)
    {
        log_[rg].clear();
        carried.clear();
        active[rg].forEach([&](uint32_t id, ActiveRec & rec) { carried.emplace_back(id, &rec); });
        std::sort(carried.begin(), carried.end(),
                  [](const std::pair<uint32_t, ActiveRec *> & a, const std::pair<uint32_t, ActiveRec *> & b) { return a.first < b.first; });
        for (auto & c : carried)
        {
            c.second->logIdx = static_cast<uint32_t>(log_[rg].size());
            log_[rg].push_back(PairInstance{c.second->dev, c.second->start, 0u, kNoLog, c.first});
        }
    }
}

std::vector<std::vector<PairInstance>> & ChromProfile::endSegmentLog()
{
    for (size_t rg = 0; rg < active.size(); ++rg)
    {
        active[rg].forEach([](uint32_t, ActiveRec & rec) { rec.logIdx = kNoLog; });
        std::vector<PairInstance> & v = log_[rg];
        size_t keep = 0;
        for (size_t i = 0; i < v.size(); ++i)
        {
            if (v[i].off == kNoLog)
                v[i].off = logWindow_;
            if (v[i].on < v[i].off)
                v[keep++] = v[i];
        }
        v.resize(keep);
    }
    logging_ = false;
    return log_;
}

// ----------------------------------------------------------------------------------------------------------
// The window loop in closed form
// ----------------------------------------------------------------------------------------------------------
bool ChromProfile::canPlanSegment(unsigned shift) const
{
    for (const StartSlots & st : starts)
        if (!st.slots[st.readSlot].sorted)
            return false;
    if (!profilesAtEnd)
        return activeKnownEmpty_;           // first segment: its set-up rule asks which read groups have no active pair
    // the extra step must succeed: else the segment has no window, yet its set-up may activate pairs
    if (currentPos + shift >= currentRightBorder)
        return false;
    for (size_t rg = 0; rg < starts.size(); ++rg)
        if (!(ends[rg].readAtEnd() && starts[rg].readAtEnd()))
            return true;
    return false;
}

WalkShape ChromProfile::planSegment(unsigned shift, std::vector<uint8_t> & equalRun, std::vector<uint8_t> & switches)
{
    WalkShape sh;
    const size_t nRg = starts.size();
    equalRun.assign(nRg, 0);
    switches = switches_;
    std::fill(switches_.begin(), switches_.end(), static_cast<uint8_t>(0));
    // ---- nextWindow(): the loader's step ahead of window 0 (canPlanSegment() made sure that it succeeds)
    if (profilesAtEnd)
    {
        currentPos += shift;
        sh.hasStep = true;
        sh.stepPos = currentPos;
        // (read group by read group on its own tables: in parallel -- 4,000 of them took 9 ms of every round one after another)
        const unsigned at = currentPos;
        parallelFor(nRg, [&](size_t rg) {
            StartSlots & st = starts[rg];
            EndSlots & en = ends[rg];
            st.cursorUpTo(at);
            en.advanceReadTo(at, shift);
        }, 128);
        profilesAtEnd = false;
    }
    // ---- initializeActive(): window 0 sits at the smallest first start, if that lies before the current position
    for (size_t rg = 0; rg < nRg; ++rg)
    {
        StartSlots & st = starts[rg];
        st.cursorToStart();
        if (!st.readAtEnd() && st.firstStart() < currentPos)
            currentPos = st.firstStart();
    }
    globalMinPos = currentPos;
    if (!sh.hasStep)
    {
        // first segment after a reset (no pair is active anywhere): the run of entries on the very first position is active
        // in window 0. (With the extra step the rule changes nothing: such entries were activated by the step.)
        for (size_t rg = 0; rg < nRg; ++rg)
        {
            StartSlots & st = starts[rg];
            if (st.readAtEnd() || st.firstStart() != currentPos)
                continue;
            equalRun[rg] = 1;
            st.cursorUpTo(currentPos);          // (behind the run of entries on that position: the slot is in start order)
        }
    }
    sh.firstPos = currentPos;
    // ---- walkSegment(): per read group the step at which both of its cursors are through, capped by the slot border
    const unsigned pos0 = currentPos;
    unsigned maxSteps = 0;
    if (pos0 + shift < currentRightBorder)
        maxSteps = (currentRightBorder - pos0 - 1) / shift;
    std::atomic<unsigned> stepsSeen(0);
    parallelFor(nRg, [&](size_t rg) {
        StartSlots & st = starts[rg];
        EndSlots & en = ends[rg];
        if (maxSteps == 0 || (en.readAtEnd() && st.readAtEnd()))
            return;
        unsigned js = 0, je = 0;
        if (!st.readAtEnd())
        {
            const unsigned last = st.slots[st.readSlot].backPos;
            js = last <= pos0 ? 1u : (last - pos0 + shift - 1) / shift;
        }
        if (!en.readAtEnd())
        {
            const unsigned last = en.lastWindowOfReadSlot() + shift;
            je = last < pos0 ? 1u : (last - pos0) / shift + 1;
        }
        const unsigned mine = std::min(maxSteps, std::max(js, je));
        unsigned seen = stepsSeen.load(std::memory_order_relaxed);
        while (seen < mine && !stepsSeen.compare_exchange_weak(seen, mine, std::memory_order_relaxed))
        {
        }
        const unsigned lastPos = pos0 + mine * shift;
        st.cursorUpTo(lastPos);
        en.advanceReadTo(lastPos, shift);
    }, 128);
    const unsigned steps = stepsSeen.load();
    currentPos = pos0 + steps * shift;
    profilesAtEnd = true;
    logWindow_ = steps + 1;
    sh.nWindows = steps + 1;
    activeKnownEmpty_ = false;
    return sh;
}

ChromProfile::PlanState ChromProfile::savePlanState() const
{
    PlanState s{currentPos, globalMinPos, logWindow_, profilesAtEnd, activeKnownEmpty_, switches_, {}};
    for (size_t rg = 0; rg < starts.size(); ++rg)
        s.cursors.emplace_back(starts[rg].readIdx, ends[rg].readIdx);
    return s;
}

void ChromProfile::restorePlanState(const PlanState & s)
{
    currentPos = s.currentPos;
    globalMinPos = s.globalMinPos;
    logWindow_ = s.logWindow;
    profilesAtEnd = s.profilesAtEnd;
    activeKnownEmpty_ = s.activeKnownEmpty;
    switches_ = s.switches;
    for (size_t rg = 0; rg < starts.size(); ++rg)
    {
        starts[rg].readIdx = s.cursors[rg].first;
        ends[rg].readIdx = s.cursors[rg].second;
    }
}

size_t WalkReference::activeTotal() const
{
    size_t n = 0;
    for (const auto & v : active_)
        n += v.size();
    return n;
}

void WalkReference::segment(unsigned rg, const WalkShape & shape, unsigned shift, unsigned switches, bool equalRun,
                            const StartSlots::Slot * slot, unsigned median, unsigned readLength, std::vector<PairInstance> & out)
{
    std::vector<Pair> & act = active_[rg];
    std::vector<Pair> & pend = pending_[rg];
    // ---- what the loader's slot switches did since the last segment: a switch drains the end read slot (pairs whose end
    // record sat there are gone; pairs filed ahead now sit in the read slot) and drops the start entries no window reached
    if (switches != 0)
    {
        std::vector<Pair> keep;
        if (switches == 1)
            for (Pair p : act)
                if (p.next)
                {
                    p.next = 0;
                    keep.push_back(p);
                }
        act.swap(keep);
        pend.clear();
        if (slot)
            for (size_t i = 0; i < slot->entries.size(); ++i)
            {
                const StartEntry & e = slot->entries[i];
                const int inner = e.deviation + static_cast<int>(median) - 2 * static_cast<int>(readLength);
                const uint32_t endPos = e.startPos + (inner > 0 ? static_cast<uint32_t>(inner) : 0u);
                const uint32_t lastWin = posToWin(endPos, 30);
                pend.push_back(Pair{e.startPos, e.deviation, slot->idOffset + static_cast<uint32_t>(i), lastWin,
                                    static_cast<uint8_t>(lastWin >= slot->endBorder ? 1 : 0)});
            }
    }
    const unsigned n = shape.nWindows, F = shape.firstPos;
    // window index at which a pair whose end record sits in the read slot leaves; >= n: not within this segment
    auto offOf = [&](const Pair & p) -> unsigned {
        if (p.next)
            return n;
        const unsigned over = p.lastWin + shift;                 // last position at which the pair is still active
        if (shape.hasStep && over < shape.stepPos)
            return 0;
        const unsigned j = over < F ? 1u : (over - F) / shift + 1;
        return std::min(j, n);
    };
    std::vector<Pair> stillActive, stillPending;
    for (const Pair & p : act)
    {
        const unsigned off = offOf(p);
        if (off > 0)
            out.push_back(PairInstance{p.dev, p.start, 0u, off, p.id});
        if (off >= n)
            stillActive.push_back(p);
    }
    bool inRun = equalRun;
    for (const Pair & p : pend)
    {
        unsigned on;
        if (shape.hasStep && p.start <= shape.stepPos)
            on = 0;
        else if (inRun && p.start == F)
            on = 0;
        else
            on = p.start <= F ? 1u : (p.start - F + shift - 1) / shift;
        if (p.start != F)
            inRun = false;
        if (on >= n)
        {
            stillPending.push_back(p);
            continue;
        }
        const unsigned off = std::max(offOf(p), on);          // (the walk activates before it deactivates within a step)
        if (on < off)
            out.push_back(PairInstance{p.dev, p.start, on, off, p.id});
        if (off >= n)
            stillActive.push_back(p);
    }
    act.swap(stillActive);
    pend.swap(stillPending);
}

}  // namespace pdh
