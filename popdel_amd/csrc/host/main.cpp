// `popdel` command-line entry point of the MI355X build. Only the `call` sub-command exists here; `profile` and
// `view` are out of scope for this implementation (reference popdel.cpp:15-54 dispatches all three).
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "call_driver.h"

int main(int argc, const char ** argv)
{
    if (argc < 2 || std::strcmp(argv[1], "--help") == 0 || std::strcmp(argv[1], "-h") == 0)
    {
        std::cerr << "PopDel - Population-wide deletion calling (MI355X joint-calling build)\n\n"
                     "SYNOPSIS\n    popdel call [OPTIONS] PROFILE-LIST-FILE | PROFILE1 PROFILE2 ...\n\n"
                     "Only the `call` command is provided by this build; use the reference PopDel for `profile` and `view`.\n";
        return argc < 2 ? 1 : 0;
    }
    if (std::strcmp(argv[1], "call") == 0)
        return pdh::runPopdelCall(argc - 1, argv + 1, nullptr, getenv("POPDEL_KEEP_TEARDOWN") == nullptr);
    std::cerr << "popdel: unknown or unsupported command '" << argv[1] << "' (this build provides `call` only)." << std::endl;
    return 1;
}
