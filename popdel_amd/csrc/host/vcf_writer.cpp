#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <sstream>
#include <string>
#include <vector>

#include "parallel.h"
#include "profile_io.h"
#include "window_call.h"

#ifndef POPDEL_VERSION
#define POPDEL_VERSION "1.5.0"
#endif

namespace pdh {
namespace {

unsigned genotypeQuality(const std::array<uint32_t, 3> & pl)
{
    if (pl[0] + pl[1] + pl[2] == 0u)
        return 0;
    if (pl[0] == 0u)
        return std::min(pl[1], pl[2]);
    if (pl[1] == 0u)
        return std::min(pl[0], pl[2]);
    return std::min(pl[0], pl[1]);
}

// Decimal digits of an unsigned value written straight into a buffer (a record has 13 numbers per sample column; at
// 10,000+ samples the formatter is the next host cost after the loader). Returns the position behind the last digit.
inline char * putUInt(char * p, unsigned v)
{
    char tmp[10];
    int n = 0;
    do
    {
        tmp[n++] = static_cast<char>('0' + v % 10u);
        v /= 10u;
    } while (v != 0u);
    while (n > 0)
        *p++ = tmp[--n];
    return p;
}

void appendSampleColumn(std::string & out, const WindowCall & c, size_t s)
{
    char buf[176];                                   // 13 numbers of at most 10 digits + separators
    char * p = buf;
    const auto & pl = c.pl[s];
    const unsigned gq = std::min(genotypeQuality(pl), 255u);
    if (gq == 0u)
    {
        std::memcpy(p, "./.:0,0,0", 9);
        p += 9;
    }
    else
    {
        *p++ = pl[2] == 0 ? '1' : '0';
        *p++ = '/';
        *p++ = pl[0] != 0 ? '1' : '0';
        *p++ = ':';
        p = putUInt(p, std::min(pl[0], 255u));
        *p++ = ',';
        p = putUInt(p, std::min(pl[1], 255u));
        *p++ = ',';
        p = putUInt(p, std::min(pl[2], 255u));
    }
    const auto & lad = c.lad[s];
    const auto & dad = c.dad[s];
    const auto & fl = c.firstLast[s];
    *p++ = ':';
    p = putUInt(p, gq);
    *p++ = ':';
    p = putUInt(p, lad[0]);
    *p++ = ',';
    p = putUInt(p, lad[1]);
    *p++ = ',';
    p = putUInt(p, lad[2]);
    *p++ = ':';
    for (int k = 0; k < 5; ++k)
    {
        if (k)
            *p++ = ',';
        p = putUInt(p, dad[k]);
    }
    *p++ = ':';
    p = putUInt(p, fl[0]);
    *p++ = ',';
    p = putUInt(p, fl[1]);
    *p++ = ':';
    p = putUInt(p, fl[1] - fl[0]);
    out.append(buf, static_cast<size_t>(p - buf));
}

std::string filterText(const WindowCall & c)
{
    if (passesAll(c))
        return "PASS";
    std::string f;
    auto add = [&](const char * name) {
        if (!f.empty())
            f.push_back(';');
        f += name;
    };
    if (c.filter & kFilterLR)
        add("lowLR");
    if (c.filter & kFilterCoverage)
        add("highCov");
    if (c.filter & kFilterSamples)
        add("missingSamples");
    if (c.filter & kFilterGT0)
        add("allRefGT");
    if (c.filter & kFilterRelWinCov)
        add("CSWin");
    return f;
}

}  // namespace

std::string formatRecord(const QuantileTable & q, const WindowCall & call, const std::string & contig)
{
    std::string line;
    line.reserve(160 + call.pl.size() * 64);
    unsigned beginPos = call.position > 1 ? call.position - 1 : call.position;
    double errorProb = 1 - q.probability(call.likelihoodRatio);
    float qual = errorProb == 0 ? 100.0f : static_cast<float>(roundHalfUp(-10 * std::log10(errorProb)));
    char buf[64];
    line += contig;
    line.push_back('\t');
    snprintf(buf, sizeof(buf), "%u", beginPos + 1);
    line += buf;
    line += "\t.\tN\t<DEL>\t";
    snprintf(buf, sizeof(buf), "%g", static_cast<double>(qual));
    line += buf;
    line.push_back('\t');
    line += filterText(call);
    line.push_back('\t');
    unsigned n = static_cast<unsigned>(call.pl.size());
    double absYield = n;
    for (const auto & p : call.pl)
        if (p[0] + p[1] + p[2] == 0u)
            --absYield;
    std::ostringstream info;
    info << "IMPRECISE;"
         << "SVLEN=" << -static_cast<int>(call.deletionLength) << ";"
         << "END=" << call.position + call.deletionLength << ";"
         << "SVTYPE=DEL;"
         << "AF=" << call.frequency << ";"
         << "LR=" << call.likelihoodRatio << ";"
         << "SVMETHOD=PopDelv" << POPDEL_VERSION << ";"
         << "YIELD=" << absYield / n << ";"
         << "SWIN=" << call.significantWindows;
    line += info.str();
    line += "\tGT:PL:GQ:LAD:DAD:FL:FLD";
    // the sample columns: in blocks on the host threads for large cohorts, joined in order
    const size_t S = call.pl.size(), kBlock = 2048;
    if (S < 4 * kBlock)
    {
        for (size_t s = 0; s < S; ++s)
        {
            line.push_back('\t');
            appendSampleColumn(line, call, s);
        }
        return line;
    }
    std::vector<std::string> parts((S + kBlock - 1) / kBlock);
    parallelFor(parts.size(), [&](size_t b) {
        std::string & out = parts[b];
        out.reserve(kBlock * 48);
        for (size_t s = b * kBlock; s < std::min(S, (b + 1) * kBlock); ++s)
        {
            out.push_back('\t');
            appendSampleColumn(out, call, s);
        }
    });
    for (const std::string & part : parts)
        line += part;
    return line;
}

VcfOutput::VcfOutput(const std::string & outfile, const std::vector<std::string> & inputFiles,
                     const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths)
{
    out_.open(outfile, std::ios::out);
    if (!out_.good())
        throw ProfileError("[PopDel] Could not open file '" + outfile + "' for writing.");
    time_t now = time(nullptr);
    char date[24];
    date[0] = '\0';
    strftime(date, sizeof(date), "[%Y%m%d] ", localtime(&now));
    out_ << "##fileformat=VCFv4.3\n"
         << "##fileDate=" << date << "\n"
         << "##source=PopDel-V" << POPDEL_VERSION << "\n";
    for (size_t i = 0; i < contigNames.size(); ++i)
        out_ << "##contig=<ID=" << contigNames[i] << ",length=" << contigLengths[i] << ">\n";
    out_ << "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
            "##INFO=<ID=IMPRECISE,Number=0,Type=Flag,Description=\"Imprecise structural variation\">\n"
            "##INFO=<ID=SVLEN,Number=.,Type=Integer,Description=\"Difference in length between REF and ALT alleles\">\n"
            "##INFO=<ID=SVTYPE,Number=1,Type=String,Description=\"Type of structural variant\">\n"
            "##INFO=<ID=END,Number=1,Type=Integer,Description=\"End position of the structural variant\">\n"
            "##INFO=<ID=SVMETHOD,Number=1,Type=String,Description=\"Approach used to detect the structural variant\">\n"
            "##INFO=<ID=LR,Number=1,Type=String,Description=\"Log-Likelihood ratio that the test is correct\">\n"
            "##INFO=<ID=YIELD,Number=1,Type=Float,Description=\"Fraction of genotyped samples\">\n"
            "##INFO=<ID=SWIN,Number=1,Type=Integer,Description=\"Number of significant windows merged into this variant\">\n"
            "##FILTER=<ID=LowLR,Description=\"Likelihood ratio below threshold\">\n"
            "##FILTER=<ID=missingSamples,Description=\"Too many samples not genotyped\">\n"
            "##FILTER=<ID=allRefGT,Description=\"All samples genotyped as homozygous reference\">\n"
            "##FILTER=<ID=CSWin,Description=\"Low fraction of significant windows\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
            "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Phred-scaled genotype likelihoods rounded to the closest integer\">\n"
            "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality. Difference of the best and second-best PL\">\n"
            "##FORMAT=<ID=LAD,Number=3,Type=Integer,Description=\"Likelihood derived allelic depth: Count of read-pairs supporting REF, ambiguous, ALT\">\n"
            "##FORMAT=<ID=DAD,Number=5,Type=Integer,Description=\"Distribution derived allelic depth: Count of read-pairs supporting REF only, REF and ALT, neither(between the histograms), ALT only, bigger than ALT\">\n"
            "##FORMAT=<ID=FL,Number=2,Type=Integer,Description=\"Window of the first and last read active in this window\">\n"
            "##FORMAT=<ID=FLD,Number=1,Type=Integer,Description=\"Distance between first and last window\">\n";
    out_ << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
    for (const std::string & f : inputFiles)
        out_ << '\t' << sampleNameOf(f);
    out_ << '\n';
    out_.flush();
}

void VcfOutput::appendContigName(const std::string & name)
{
    if (!locked_)
    {
        contigNames_.push_back(name);
        locked_ = true;
    }
}

void VcfOutput::unlockContigName()
{
    if (locked_)
    {
        ++rID_;
        locked_ = false;
    }
}

void VcfOutput::writeCalls(const std::vector<WindowCall> & calls, const CallParams & params)
{
    std::vector<const WindowCall *> wanted;
    for (const WindowCall & c : calls)
        if (c.iterations != 0 && (params.outputFailed || passesAll(c)))
            wanted.push_back(&c);
    // records are formatted on the host threads (large records split their sample columns themselves), written in order
    std::vector<std::string> lines(wanted.size());
    const bool perRecord = !wanted.empty() && wanted[0]->pl.size() < 8192;
    if (perRecord)
        parallelFor(wanted.size(), [&](size_t i) { lines[i] = formatRecord(params.quantileTable, *wanted[i], contigNames_[rID_]); });
    else
        for (size_t i = 0; i < wanted.size(); ++i)
            lines[i] = formatRecord(params.quantileTable, *wanted[i], contigNames_[rID_]);
    for (const std::string & l : lines)
        out_ << l << '\n';
}

}  // namespace pdh
