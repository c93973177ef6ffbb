#include "call_params.h"
#include "parallel.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <unordered_set>

#include "profile_io.h"

#ifndef POPDEL_VERSION
#define POPDEL_VERSION "1.5.0"
#endif

namespace pdh {

void printStatus(const std::string & msg)
{
    time_t now = time(nullptr);
    struct tm tmv = *localtime(&now);
    char buf[80];
    strftime(buf, sizeof(buf), "[PopDel %Y-%m-%d %X] ", &tmv);
    std::cout << buf << msg << std::endl;
}

int roundHalfUp(double d) { return static_cast<int>(std::floor(d + 0.5)); }

// File name without directory and without the last extension (reference utils_popdel.h:1469-1489).
std::string sampleNameOf(const std::string & path)
{
    size_t lastDelim = 0, lastDot = path.size();
    for (size_t i = 0; i < path.size(); ++i)
    {
        if (path[i] == '/' || path[i] == '\\')
            lastDelim = i;
        else if (path[i] == '.')
            lastDot = i;
    }
    if (lastDelim == 0)
        return path.substr(0, lastDot);
    if (lastDot < lastDelim)
        return path.substr(lastDelim + 1);
    return path.substr(lastDelim + 1, lastDot - lastDelim - 1);
}

// ----------------------------------------------------------------------------------------------------------
// QUAL table: chi-square(df=1) quantiles for phred 0..100, shifted by the log prior odds
// (values: reference parameter_parsing_popdel_call.h:24-129; generated there with R's qchisq).
// ----------------------------------------------------------------------------------------------------------
QuantileTable::QuantileTable(double prior)
{
    static const double chi[101] = {
        0, 0.0679615365255401, 0.230764766661193, 0.452421556097698, 0.714036104152315, 1.00448471501902, 1.31668063342863,
        1.64583886397469, 1.98858107774449, 2.34243609502603, 2.70554345409542, 3.07646857913928, 3.45408274293754,
        3.83748240184564, 4.22593339019369, 4.61883133337202, 5.01567294625387, 5.41603482049954, 5.81955747770787,
        6.22593319775977, 6.63489660102121, 7.04621727098416, 7.45969391025114, 7.87514966368955, 8.29242834051146,
        8.71139133617301, 9.13191510451069, 9.55388906647803, 9.97721386826398, 10.4017999212068, 10.8275661706627,
        11.2544390521783, 11.6823516018745, 12.1112426945654, 12.5410563882771, 12.9717413578738, 13.403250403671,
        13.8355400234795, 14.2685700385079, 14.7023032652319, 15.1367052266236, 15.571743897219, 16.007389477407,
        16.4436141930027, 16.8803921167791, 17.317699009203, 17.7555121758597, 18.193810339586, 18.632573525518,
        19.0717829575225, 19.5114209646663, 19.9514708965797, 20.3919170468123, 20.8327445832119, 21.2739394844285,
        21.7154884822888, 22.1573790090714, 22.599599149316, 23.0421375954437, 23.4849836077724, 23.9281269768795,
        24.3715579896983, 24.8152673978314, 25.2592463886412, 25.7034865580547, 26.1479798870853, 26.5927187165227,
        27.0376957272151, 27.4829039228708, 27.9283366047418, 28.3739873627981, 28.8198500523812, 29.2659187856725,
        29.7121879123161, 30.1586520172268, 30.6053058897649, 31.0521445329994, 31.4991631422743, 31.9463570799056,
        32.3937219226572, 32.8412533514689, 33.2889472854096, 33.7367997370742, 34.1848068388937, 34.6329649593365,
        35.0812704794425, 35.5297199531207, 35.9783101801565, 36.4270377410164, 36.8758997410294, 37.3248931065187,
        37.7740147322013, 38.2232620681608, 38.672632697292, 39.1221230144588, 39.5717319767953, 40.0214554741213,
        40.4712914238083, 40.9212389675878, 41.371295018385, 41.8214562029826};
    static const double prob[101] = {
        0, 0.205671765275719, 0.369042655519807, 0.498812766372728, 0.601892829446503, 0.683772233983162, 0.748811356849042,
        0.800473768503112, 0.841510680753889, 0.874107458820583, 0.9, 0.920567176527572, 0.936904265551981, 0.949881276637273,
        0.96018928294465, 0.968377223398316, 0.974881135684904, 0.980047376850311, 0.984151068075389, 0.987410745882058, 0.99,
        0.992056717652757, 0.993690426555198, 0.994988127663727, 0.996018928294465, 0.996837722339832, 0.99748811356849,
        0.998004737685031, 0.998415106807539, 0.998741074588206, 0.999, 0.999205671765276, 0.99936904265552, 0.999498812766373,
        0.999601892829446, 0.999683772233983, 0.999748811356849, 0.999800473768503, 0.999841510680754, 0.999874107458821,
        0.9999, 0.999920567176528, 0.999936904265552, 0.999949881276637, 0.999960189282945, 0.999968377223398,
        0.999974881135685, 0.99998004737685, 0.999984151068075, 0.999987410745882, 0.99999, 0.999992056717653,
        0.999993690426555, 0.999994988127664, 0.999996018928294, 0.99999683772234, 0.999997488113568, 0.999998004737685,
        0.999998415106808, 0.999998741074588, 0.999999, 0.999999205671765, 0.999999369042656, 0.999999498812766,
        0.999999601892829, 0.999999683772234, 0.999999748811357, 0.999999800473768, 0.999999841510681, 0.999999874107459,
        0.9999999, 0.999999920567177, 0.999999936904266, 0.999999949881277, 0.999999960189283, 0.999999968377223,
        0.999999974881136, 0.999999980047377, 0.999999984151068, 0.999999987410746, 0.99999999, 0.999999992056718,
        0.999999993690427, 0.999999994988128, 0.999999996018928, 0.999999996837722, 0.999999997488114, 0.999999998004738,
        0.999999998415107, 0.999999998741075, 0.999999999, 0.999999999205672, 0.999999999369043, 0.999999999498813,
        0.999999999601893, 0.999999999683772, 0.999999999748811, 0.999999999800474, 0.999999999841511, 0.999999999874107,
        0.9999999999};
    const double p = std::log(prior / (1 - prior));
    m_.emplace(0.0, 0.0);                       // the first key is a plain 0, not shifted
    for (int i = 1; i <= 100; ++i)
        m_.emplace(chi[i] / 2 - p, prob[i]);
}

double QuantileTable::probability(double lr) const
{
    auto it = m_.upper_bound(lr);
    if (it == m_.end())
        return 1.0;
    --it;
    return it->second;
}

// ----------------------------------------------------------------------------------------------------------
// Region strings: name | name:start | name:start-end (1-based, closed); ':' may be part of the name.
// Reference: seqan/seq_io/genomic_region.h:241-282, utils_popdel.h:797-873.
// ----------------------------------------------------------------------------------------------------------
namespace {

bool parsePlainRegion(Region & reg, const std::string & s)
{
    reg = Region();
    size_t colon = s.find(':');
    reg.seqName = s.substr(0, colon);
    if (colon == std::string::npos)
        return true;
    std::string rest = s.substr(colon + 1);
    std::string a, b;
    size_t i = 0;
    for (; i < rest.size() && rest[i] != '-'; ++i)
        if (rest[i] != ',')
            a.push_back(rest[i]);
    bool hasEnd = i < rest.size();
    for (++i; i < rest.size(); ++i)
        if (rest[i] != ',')
            b.push_back(rest[i]);
    auto toInt = [](const std::string & t, int32_t & out) {
        if (t.empty())
            return false;
        char * e = nullptr;
        long v = std::strtol(t.c_str(), &e, 10);
        if (*e != '\0')
            return false;
        out = static_cast<int32_t>(v);
        return true;
    };
    if (!toInt(a, reg.beginPos) || reg.beginPos < 1)
        return false;
    --reg.beginPos;
    if (!hasEnd)
    {
        reg.endPos = -1;
        return true;
    }
    if (!toInt(b, reg.endPos) || reg.endPos < 1)
        return false;
    return true;
}

}  // namespace

bool parseRegion(Region & reg, const std::string & s)
{
    auto fail = [&]() {
        printStatus("[PopDel] ERROR: Invalid genomic region '" + s + "'.");
        return false;
    };
    if (s.empty())
    {
        printStatus("[PopDel] ERROR: Empty genomic region.");
        return false;
    }
    if (parsePlainRegion(reg, s))
    {
        if (reg.seqName.empty() || !isalnum(static_cast<unsigned char>(reg.seqName.back())))
            return fail();
        return true;
    }
    // Retry with everything up to the LAST ':' as the name.
    reg = Region();
    size_t lastCol = s.rfind(':');
    if (lastCol == std::string::npos)
        lastCol = 0;
    if (lastCol + 1 == s.size())
        return fail();
    if (!isdigit(static_cast<unsigned char>(s[lastCol + 1])))
    {
        if (!isalnum(static_cast<unsigned char>(s.back())))
            return fail();
        reg.seqName = s;
        return true;
    }
    std::string first = s.substr(0, lastCol);
    Region tmp;
    if (!parsePlainRegion(tmp, "x:" + s.substr(lastCol + 1)))
        return fail();
    if (first.empty() || !isalnum(static_cast<unsigned char>(first.back())))
        return fail();
    reg = tmp;
    reg.seqName = first;
    return true;
}

// ----------------------------------------------------------------------------------------------------------
// Command line
// ----------------------------------------------------------------------------------------------------------
namespace {

struct OptSpec
{
    char shortName;
    const char * longName;
    bool takesValue;
};

const OptSpec kOptions[] = {
    {'h', "help", false},      {'H', "fullHelp", false},          {'A', "active-coverage-file", true},
    {'a', "active-coverage", true}, {'d', "max-deletion-size", true}, {'e', "per-sample-rgid", false},
    {'l', "min-init-length", true}, {'m', "min-length", true},     {'o', "out", true},
    {'r', "region-of-interest", true}, {'R', "ROI-file", true},    {'s', "min-sample-fraction", true},
    {'C', "cell-population", false}, {'b', "buffer-size", true},   {'c', "min-relative-window-cover", true},
    {'f', "pseudocount-fraction", true}, {'F', "output-failed", false}, {'n', "no-regenotyping", false},
    {'p', "prior-probability", true}, {'t', "iterations", true},   {'u', "unsmoothed", false},
    {'x', "uncompressed-in", false}, {0, "version", false}};

void printCallHelp(bool full)
{
    std::cout <<
        "PopDel call - Population-wide deletion calling (MI355X build)\n"
        "=============================================================\n\n"
        "SYNOPSIS\n"
        "    popdel call [OPTIONS] PROFILE-LIST-FILE\n"
        "    popdel call [OPTIONS] PROFILE-FILE1 PROFILE-FILE2 [PROFILE-FILE3 ...]\n\n"
        "DESCRIPTION\n"
        "    Performs joint-calling of deletions using a list of profile-files previously created using the\n"
        "    'popdel profile' command. The input profiles are either specified directly as arguments or listed in\n"
        "    PROFILE-LIST-FILE (one filename per line).\n\n"
        "OPTIONS\n"
        "    -h, --help                          Display the help message.\n"
        "    -H, --fullHelp                      Displays full list of options.\n"
        "    -A, --active-coverage-file FILE     File with lines consisting of \"ReadGroup  maxCov\".\n"
        "    -a, --active-coverage NUM           Maximum number of active read pairs (~coverage). Default: 100.\n"
        "    -d, --max-deletion-size NUM         Maximum size of deletions. Default: 10000.\n"
        "    -e, --per-sample-rgid               Internally modify each read group ID by adding the filename.\n"
        "    -l, --min-init-length NUM           Minimal deletion length at initialization of iteration.\n"
        "    -m, --min-length NUM                Minimal deletion length during iteration.\n"
        "    -o, --out FILE                      Output file name. Default: popdel.vcf.\n"
        "    -r, --region-of-interest REGION     Genomic region 'chr:start-end' (closed interval, 1-based index).\n"
        "    -R, --ROI-file FILE                 File listing one or more regions of interest, one region per line.\n"
        "    -s, --min-sample-fraction NUM       Minimum fraction of samples required to have enough data. Default: 0.1.\n"
        "    -C, --cell-population               Use genotype weights for cell population instead of human population.\n";
    if (full)
        std::cout <<
            "    -b, --buffer-size NUM               Number of buffered windows. Default: 200000.\n"
            "    -c, --min-relative-window-cover NUM Fraction of a deletion that has to be covered by significant windows. Default: 0.5.\n"
            "    -f, --pseudocount-fraction NUM      Divisor of the biggest background likelihood for the pseudocounts. Default: 500.\n"
            "    -F, --output-failed                 Also output calls which did not pass the filters.\n"
            "    -n, --no-regenotyping               Outputs every potential variant window without re-genotyping and merging.\n"
            "    -p, --prior-probability NUM         Prior probability of a deletion. Default: 0.0001.\n"
            "    -t, --iterations NUM                Number of iterations in EM for length estimation. Default: 15.\n"
            "    -u, --unsmoothed                    Disable the smoothing of the insert size histogram.\n"
            "    -x, --uncompressed-in               Read uncompressed PopDel profiles.\n";
    std::cout << "\nVERSION\n    PopDel version: " << POPDEL_VERSION << "\n";
}

bool toUnsigned(const std::string & s, unsigned & out)
{
    if (s.empty() || s[0] == '-')
        return false;
    char * e = nullptr;
    unsigned long v = std::strtoul(s.c_str(), &e, 10);
    if (*e != '\0')
        return false;
    out = static_cast<unsigned>(v);
    return true;
}

bool toDouble(const std::string & s, double & out)
{
    if (s.empty())
        return false;
    char * e = nullptr;
    out = std::strtod(s.c_str(), &e);
    return *e == '\0';
}

// Read file names from a list file, dropping duplicates (reference utils_popdel.h:879-912).
void loadFilenames(std::vector<std::string> & files)
{
    std::string listFile = files[0];
    files.clear();
    std::ifstream in(listFile);
    if (!in.is_open())
        throw ProfileError("[PopDel] Could not open file '" + listFile + "' for reading.");
    std::unordered_set<std::string> seen;
    std::string name;
    while (in >> name)
    {
        if (seen.insert(name).second)
            files.push_back(name);
        else
            printStatus("WARNING: duplicate file " + name + ". Ignoring additional occurences.");
    }
    std::ostringstream msg;
    msg << "Loaded " << files.size() << " filenames from '" << listFile << "'.";
    printStatus(msg);
}

}  // namespace

int parseCallCommandLine(CallParams & params, int argc, const char ** argv)
{
    std::map<std::string, std::vector<std::string>> values;   // by long name
    std::vector<std::string> positional;
    auto err = [](const std::string & m) {
        std::cerr << "popdel call: " << m << std::endl;
        return 1;
    };
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i];
        const OptSpec * spec = nullptr;
        std::string inlineValue;
        bool hasInline = false;
        if (a.size() > 2 && a[0] == '-' && a[1] == '-')
        {
            std::string name = a.substr(2);
            size_t eq = name.find('=');
            if (eq != std::string::npos)
            {
                inlineValue = name.substr(eq + 1);
                name = name.substr(0, eq);
                hasInline = true;
            }
            for (const OptSpec & o : kOptions)
                if (name == o.longName || (name == "full-help" && o.shortName == 'H'))
                    spec = &o;
            if (!spec)
                return err("illegal option -- " + a);
        }
        else if (a.size() >= 2 && a[0] == '-' && !isdigit(static_cast<unsigned char>(a[1])))
        {
            for (const OptSpec & o : kOptions)
                if (o.shortName != 0 && a[1] == o.shortName)
                    spec = &o;
            if (!spec)
                return err("illegal option -- " + a);
            if (a.size() > 2)
            {
                if (spec->takesValue)
                {
                    inlineValue = a.substr(2);
                    hasInline = true;
                }
                else
                {
                    // bundle of flags, e.g. -nF
                    for (size_t k = 1; k < a.size(); ++k)
                    {
                        const OptSpec * s2 = nullptr;
                        for (const OptSpec & o : kOptions)
                            if (o.shortName != 0 && a[k] == o.shortName && !o.takesValue)
                                s2 = &o;
                        if (!s2)
                            return err("illegal option -- " + a);
                        values[s2->longName].push_back("");
                    }
                    continue;
                }
            }
        }
        else
        {
            positional.push_back(a);
            continue;
        }
        if (spec->takesValue)
        {
            if (!hasInline)
            {
                if (i + 1 >= argc)
                    return err(std::string("option requires an argument -- ") + spec->longName);
                inlineValue = argv[++i];
            }
            values[spec->longName].push_back(inlineValue);
        }
        else
        {
            values[spec->longName].push_back("");
        }
    }
    if (values.count("fullHelp"))
    {
        printCallHelp(true);
        return 0;
    }
    if (values.count("help"))
    {
        printCallHelp(false);
        return 0;
    }
    if (values.count("version"))
    {
        std::cout << "PopDel version: " << POPDEL_VERSION << std::endl;
        return 0;
    }
    if (positional.empty())
        return err("Not enough arguments were provided.");

    auto last = [&](const char * n) -> const std::string & { return values[n].back(); };
    auto uintOpt = [&](const char * n, unsigned & dst, long minV) -> bool {
        if (!values.count(n))
            return true;
        unsigned v;
        if (!toUnsigned(last(n), v) || static_cast<long>(v) < minV)
        {
            std::cerr << "popdel call: the given value '" << last(n) << "' is not valid for option --" << n << std::endl;
            return false;
        }
        dst = v;
        return true;
    };
    auto dblOpt = [&](const char * n, double & dst, double minV, double maxV) -> bool {
        if (!values.count(n))
            return true;
        double v;
        if (!toDouble(last(n), v) || v < minV || v > maxV)
        {
            std::cerr << "popdel call: the given value '" << last(n) << "' is not valid for option --" << n << std::endl;
            return false;
        }
        dst = v;
        return true;
    };
    if (values.count("out"))
        params.outfile = last("out");
    if (!dblOpt("prior-probability", params.prior, 0.0, 0.9999) || !uintOpt("iterations", params.iterations, 0) ||
        !dblOpt("min-sample-fraction", params.minSampleFraction, 0.0, 1.0) ||
        !uintOpt("buffer-size", params.windowBuffer, 10000) ||
        !dblOpt("min-relative-window-cover", params.minRelWinCover, 0.0, 2.0) ||
        !uintOpt("active-coverage", params.defaultMaxLoad, 0) ||
        !uintOpt("pseudocount-fraction", params.pseudoCountFraction, 50) ||
        !uintOpt("max-deletion-size", params.maxDeletionSize, 0))
        return 1;
    if (values.count("ROI-file"))
        params.roiFile = last("ROI-file");
    if (values.count("active-coverage-file"))
        params.maxLoadFile = last("active-coverage-file");
    params.smoothing = !values.count("unsmoothed");
    if (values.count("min-init-length"))
    {
        unsigned v = 0;
        if (!uintOpt("min-init-length", v, 0))
            return 1;
        params.minInitDelLengths.assign(1, v);
    }
    params.windowWiseOutput = values.count("no-regenotyping") != 0;
    if (!uintOpt("min-length", params.minLen, 0))
        return 1;
    if (values.count("region-of-interest"))
        params.roiList = values["region-of-interest"];
    params.outputFailed = values.count("output-failed") != 0;
    params.modRgByFileName = values.count("per-sample-rgid") != 0;
    params.uncompressedIn = values.count("uncompressed-in") != 0;
    params.somatic = values.count("cell-population") != 0;

    // Echo the command like the reference does.
    {
        std::cout << "PopDel - Population-wide deletion calling\n=========================================\n\n";
        std::ostringstream cmd;
        cmd << "Command:  \n\n";
        for (int i = 1; i < argc; ++i)
            cmd << argv[i] << " ";
        cmd << std::endl;
        printStatus(cmd);
    }
    params.inputFiles = positional;
    if (params.inputFiles.size() == 1)
        loadFilenames(params.inputFiles);
    params.sampleNum = static_cast<unsigned>(params.inputFiles.size());
    params.fileCount = params.sampleNum;
    return -1;
}

// ----------------------------------------------------------------------------------------------------------
// Start-up calculations
// ----------------------------------------------------------------------------------------------------------
namespace {

bool registerReadGroup(std::map<std::string, unsigned> & readGroups, unsigned & rg, const std::string & name,
                       const std::string & filename)
{
    if (readGroups.count(name) != 0)
    {
        printStatus("WARNING: Duplicate insert size histogram of read group '" + name + "'. Skipping it in file '" +
                    filename + "'.");
        return false;
    }
    readGroups[name] = rg;
    ++rg;
    return true;
}

// (the header has been read and its histograms conditioned by loadAndCalculateParameters(): in parallel over the files; what
// follows is the order-dependent part -- read-group registration, the checks and their messages -- file by file)
void loadProfileHeaderInto(CallParams & p, std::vector<unsigned> & sampleRgs, const std::string & filename, ProfileHeader & hdr)
{
    const bool dropContigs = !p.contigNames.empty() && p.representativeContigs;
    if (p.representativeContigs && !p.indexRegionSizes.empty() && hdr.indexRegionSize != p.indexRegionSizes[0])
        throw ProfileError("Index region sizes are not equal! Terminating.");
    unsigned rg = static_cast<unsigned>(p.readGroups.size());
    bool added = false;
    for (size_t i = 0; i < hdr.readGroups.size(); ++i)
    {
        if (!registerReadGroup(p.readGroups, rg, hdr.readGroups[i], filename))
        {
            if (!p.modRgByFileName)
                throw ProfileError("Previously observed read groups detected in file '" + filename +
                                   "'. Please make sure that all read group IDs of your cohort are unique or remove the "
                                   "sample(s) in question. If you are sure that the duplicate read group names are OK please "
                                   "use the option '-e'/'--per-sample-rgid' to resolve the conflict. Terminating.");
            // -e: prefix every read group of this file with the sample name and retry this one.
            for (std::string & name : hdr.readGroups)
            {
                std::string renamed = sampleNameOf(filename) + ":" + name;
                std::cout << "WARNING: Internally replacing read group ID '" << name << "' of file '" << filename
                          << "' with '" << renamed << "'" << std::endl;
                name = renamed;
            }
            if (!registerReadGroup(p.readGroups, rg, hdr.readGroups[i], filename))
                throw ProfileError("This should not have happened... Tried to de-duplicate read groups of '" + filename +
                                   "' by adding the file name to the IDs, but something went wrong. Terminating.");
        }
        sampleRgs.push_back(rg - 1);
        added = true;
        p.histograms.push_back(std::move(hdr.histograms[i]));
    }
    if (!added)
        throw ProfileError("All read group IDs of sample '" + filename + "' are duplicates of previously observed read group IDs.");
    if (!dropContigs)
    {
        p.contigNames.push_back(hdr.contigNames);
        p.contigLengths.push_back(hdr.contigLengths);
        p.indexRegionSizes.push_back(hdr.indexRegionSize);
    }
}

unsigned percentileOf(std::vector<unsigned> values, double p)
{
    std::sort(values.begin(), values.end());
    unsigned n = static_cast<unsigned>(values.size());
    unsigned j = static_cast<unsigned>(std::floor(n * p));
    bool remainder = j != n * p;
    return remainder ? values[j] : values[j - 1];
}

bool regionOrder(const Region & a, const Region & b)
{
    int c = a.seqName.compare(b.seqName);
    if (c < 0)
        return true;
    if (c > 0)
        return false;
    return a.beginPos <= b.beginPos;
}

void mergeOverlapping(std::vector<Region> & v)
{
    if (v.size() < 2)
        return;
    // Insertion sort with the reference's (non-strict) comparator keeps stable_sort semantics for equal begins.
    std::stable_sort(v.begin(), v.end(), [](const Region & a, const Region & b) {
        int c = a.seqName.compare(b.seqName);
        if (c != 0)
            return c < 0;
        return a.beginPos < b.beginPos;
    });
    (void)regionOrder;
    std::vector<Region> merged;
    merged.push_back(v[0]);
    for (size_t i = 1; i < v.size(); ++i)
    {
        Region & lastR = merged.back();
        if (lastR.seqName == v[i].seqName && lastR.endPos >= v[i].beginPos)
        {
            std::ostringstream msg;
            msg << "WARNING: Intervals on '" << lastR.seqName << "' overlap and have been merged.";
            if (lastR.endPos < v[i].endPos)
                lastR.endPos = v[i].endPos;
            printStatus(msg);
        }
        else
        {
            merged.push_back(v[i]);
        }
    }
    v.swap(merged);
}

bool initializeRois(CallParams & p, const std::vector<std::string> & intervalStrings, const std::vector<std::string> & contigOrder)
{
    std::map<std::string, std::vector<Region>> byContig;
    for (const std::string & s : intervalStrings)
    {
        Region r;
        if (!parseRegion(r, s))
            throw ProfileError("[PopDel] Error while parsing genomic region '" + s + "'. Terminating.");
        if (r.beginPos < 0)
            r.beginPos = 0;
        if (r.endPos < 0)
            r.endPos = std::numeric_limits<int32_t>::max();
        byContig[r.seqName].push_back(r);
    }
    std::vector<Region> all;
    for (auto & kv : byContig)
    {
        mergeOverlapping(kv.second);
        for (const Region & r : kv.second)
            all.push_back(r);
    }
    if (all.empty())
        return false;
    // All names must be known (reference checkRois(), load_profile_popdel_call.h:52-78).
    for (const Region & r : all)
        if (std::find(contigOrder.begin(), contigOrder.end(), r.seqName) == contigOrder.end())
            throw ProfileError("[PopDel] Invalid chromosome name '" + r.seqName +
                               "' not part of reference sequence names in BAM header. Terminating.");
    // Re-order by the first profile's contig order, ties by current position.
    std::vector<std::pair<std::pair<unsigned, unsigned>, Region>> keyed;
    for (size_t i = 0; i < all.size(); ++i)
    {
        unsigned j = static_cast<unsigned>(std::find(contigOrder.begin(), contigOrder.end(), all[i].seqName) - contigOrder.begin());
        keyed.push_back({{j, static_cast<unsigned>(i)}, all[i]});
    }
    std::sort(keyed.begin(), keyed.end(), [](const auto & a, const auto & b) { return a.first < b.first; });
    p.allRois.clear();
    for (auto & k : keyed)
        p.allRois.push_back(k.second);
    p.nextRoi = 0;
    return true;
}

void loadMaxLoad(CallParams & p)
{
    if (p.defaultMaxLoad == 0u)
        p.defaultMaxLoad = std::numeric_limits<unsigned>::max();
    p.maxLoad.assign(p.readGroups.size(), p.defaultMaxLoad);
    if (p.maxLoadFile.empty())
        return;
    std::ifstream file(p.maxLoadFile);
    if (!file.is_open())
        throw ProfileError("[PopDel] Could not open coverage file '" + p.maxLoadFile + "' for reading.");
    std::map<std::string, unsigned> loadMap;
    std::string rgName, load;
    unsigned len = 0;
    while (file >> rgName)
    {
        if (!(file >> load))
            throw ProfileError("[PopDel] Could not read maximum coverage for read group '" + rgName + "'.");
        loadMap[rgName] = static_cast<unsigned>(std::stoul(load));
        ++len;
    }
    unsigned assigned = 0;
    std::vector<std::string> unassigned;
    for (const auto & kv : p.readGroups)
    {
        auto it = loadMap.find(kv.first);
        if (it != loadMap.end())
        {
            p.maxLoad[kv.second] = it->second == 0u ? std::numeric_limits<unsigned>::max() : it->second;
            ++assigned;
        }
        else
        {
            unassigned.push_back(kv.first);
        }
    }
    std::ostringstream msg;
    msg << "Loaded maximum active coverage for " << len << " read groups. (" << assigned << " assigned)";
    printStatus(msg);
    if (!unassigned.empty())
    {
        std::ostringstream w;
        w << "WARNING: No maximum active coverage value specified for the following read groups: ";
        for (size_t i = 0; i < unassigned.size(); ++i)
            w << (i ? ", " : "") << unassigned[i];
        w << "\nAssuming the default value of " << p.defaultMaxLoad << " for the listed read group(s).";
        printStatus(w);
    }
}

}  // namespace

void loadAndCalculateParameters(CallParams & p)
{
    const size_t n = p.inputFiles.size();
    p.rgs.assign(n, std::vector<unsigned>());
    unsigned long long totalSize = 0;
    // headers read and histograms conditioned in parallel (a thousand files: 50 ms one after the other); a file that cannot
    // be read reports when its turn comes, behind the messages of the files before it
    struct Loaded
    {
        ProfileHeader hdr;
        std::exception_ptr error;
    };
    std::vector<Loaded> loaded(n);
    parallelFor(n, [&](size_t i) {
        try
        {
            ProfileFile file(p.inputFiles[i]);
            file.readHeader(loaded[i].hdr, i != 0 && p.representativeContigs);      // (contig table: the first file's, or every file's)
            for (InsertHistogram & h : loaded[i].hdr.histograms)
                conditionHistogram(h, 256, p.smoothing, p.pseudoCountFraction);
        }
        catch (...)
        {
            loaded[i].error = std::current_exception();
        }
    }, 4);
    for (size_t i = 0; i < n; ++i)
    {
        if (loaded[i].error)
            std::rethrow_exception(loaded[i].error);
        loadProfileHeaderInto(p, p.rgs[i], p.inputFiles[i], loaded[i].hdr);
        // Size report of the reference: 24-byte string object + values + 4 doubles + 1 int + 5 unsigned,
        // taken from the histogram with the FILE's index.
        const InsertHistogram & h = p.histograms[i < p.histograms.size() ? i : p.histograms.size() - 1];
        unsigned histSize = static_cast<unsigned>(24 + h.values.size() * sizeof(double) + 4 * sizeof(double) + sizeof(int) + 5 * sizeof(unsigned));
        totalSize += histSize;
        std::ostringstream msg;
        msg << "Loaded histogram from file '" << p.inputFiles[i] << "' in " << histSize << " byte.";
        printStatus(msg);
    }
    {
        std::ostringstream msg;
        msg << "Total memory reserved for histograms amounts to " << totalSize << " byte.";
        printStatus(msg);
    }
    {
        std::ostringstream msg;
        msg << "Loaded insert size histograms for " << p.histograms.size() << " read groups.";
        printStatus(msg);
    }
    if (p.windowShift == 0)
        p.windowShift = p.windowSize;
    // mean of the standard deviations
    {
        double sum = 0;
        for (const InsertHistogram & h : p.histograms)
            sum += h.stddev;
        p.meanStddev = sum / p.histograms.size();
    }
    // minimum initial deletion length per read group: user value or round(4 sigma)
    {
        std::ostringstream msg;
        if (p.minInitDelLengths.size() == 1)
        {
            unsigned v = p.minInitDelLengths[0];
            p.minInitDelLengths.assign(p.histograms.size(), v);
            msg << "The minimum initial deletion length has been set to " << v << " for all samples and readgroups.";
        }
        else
        {
            const double factor = 4;
            p.minInitDelLengths.assign(p.histograms.size(), 0);
            unsigned lo = std::numeric_limits<unsigned>::max(), hi = 0;
            for (size_t i = 0; i < p.histograms.size(); ++i)
            {
                p.minInitDelLengths[i] = static_cast<unsigned>(roundHalfUp(factor * p.histograms[i].stddev));
                lo = std::min(lo, p.minInitDelLengths[i]);
                hi = std::max(hi, p.minInitDelLengths[i]);
            }
            msg << "Minimum initial deletion lengths have been set to " << factor
                << " * standard deviations of insert size histograms [" << lo << ".." << hi << "].";
        }
        printStatus(msg);
    }
    // minimum final deletion length: 95th percentile of the initial minimum lengths
    {
        if (p.minLen == std::numeric_limits<unsigned>::max())
            p.minLen = static_cast<unsigned>(roundHalfUp(1.0 * percentileOf(p.minInitDelLengths, 0.95)));
        std::ostringstream msg;
        msg << "Minimum final deletion length has been set to " << p.minLen << ".";
        printStatus(msg);
    }
    // LR threshold: half the 99 % chi-square(1) quantile minus the log prior odds
    p.minimumLikelihoodRatio = (6.6349 / 2.0) - std::log(p.prior / (1 - p.prior));

    if (!p.roiFile.empty())
    {
        std::ifstream file(p.roiFile);
        if (!file.is_open())
            throw ProfileError("[PopDel] Could not open interval file '" + p.roiFile + "' for reading.");
        std::string word;
        unsigned len = 0;
        while (file >> word)
        {
            p.roiList.push_back(word);
            ++len;
        }
        std::ostringstream msg;
        msg << "Finished reading " << len << " intervals from file '" << p.roiFile << "'.";
        printStatus(msg);
        initializeRois(p, p.roiList, p.contigNames[0]);
    }
    else if (!p.roiList.empty())
    {
        initializeRois(p, p.roiList, p.contigNames[0]);
    }
    else
    {
        initializeRois(p, p.contigNames[0], p.contigNames[0]);
    }
    loadMaxLoad(p);
    std::ostringstream msg;
    msg << "Calculated minimum log-likelihood ratio as " << p.minimumLikelihoodRatio << " from the prior probability "
        << p.prior << " using the 99%-quantile of a chi-squared distribution with df=1.";
    printStatus(msg);
}

}  // namespace pdh
