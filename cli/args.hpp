// Command-line grammar of the stft / stft_flt tools: the option syntax the reference gets from
// argtable2 (reference src/stft.cpp:95-115): short options with the value attached or separate
// (-s65, -s 65), grouped flags (-ez), long options with = or a separate value (--step=65,
// --step 65), "-" as a positional meaning stdin, "--" ending the options.
#pragma once
#include <string>
#include <vector>

namespace dspcli {

enum class ArgKind { Flag, Int, Dbl, File };

struct ArgSpec {
    char shortname;            // 0 = none
    const char *longname;
    const char *datatype;      // "<uint>", "<dbl>", "<file>" or nullptr for flags
    ArgKind kind;
    const char *glossary;
    int count = 0;
    long long ival = 0;
    double dval = 0.0;
    std::string sval;
};

struct ArgParser {
    std::string prog;
    std::vector<ArgSpec> specs;
    std::vector<std::string> positionals;
    int max_positionals = 2;
    std::vector<std::string> errors;

    ArgSpec *find_short(char c);
    ArgSpec *find_long(const std::string &name);
    ArgSpec &operator[](const char *longname);
    bool parse(int argc, char **argv);
    void print_usage(std::ostream &os, const char *positional_help) const;
};

}  // namespace dspcli
