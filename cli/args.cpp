#include "args.hpp"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <ostream>
#include <stdexcept>

namespace dspcli {

ArgSpec *ArgParser::find_short(char c)
{
    for (auto &s : specs) { if (s.shortname == c) { return &s; } }
    return nullptr;
}

ArgSpec *ArgParser::find_long(const std::string &name)
{
    for (auto &s : specs) { if (name == s.longname) { return &s; } }
    return nullptr;
}

ArgSpec &ArgParser::operator[](const char *longname)
{
    ArgSpec *s = find_long(longname);
    if (!s) { throw std::logic_error(std::string("no such option: ") + longname); }
    return *s;
}

static bool set_value(ArgSpec &s, const std::string &text, const std::string &shown, std::vector<std::string> &errors)
{
    if (s.count > 0) { errors.push_back("excess option " + shown); return false; }
    if (s.kind == ArgKind::Int) {
        errno = 0;
        char *end = nullptr;
        const long long v = std::strtoll(text.c_str(), &end, 0);
        if (text.empty() || *end != '\0' || errno == ERANGE) { errors.push_back("invalid argument \"" + text + "\" to option " + shown); return false; }
        s.ival = v;
    } else if (s.kind == ArgKind::Dbl) {
        errno = 0;
        char *end = nullptr;
        const double v = std::strtod(text.c_str(), &end);
        if (text.empty() || *end != '\0') { errors.push_back("invalid argument \"" + text + "\" to option " + shown); return false; }
        s.dval = v;
    } else {
        s.sval = text;
    }
    ++s.count;
    return true;
}

bool ArgParser::parse(int argc, char **argv)
{
    bool options_done = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (options_done || a == "-" || a.empty() || a[0] != '-') {
            if ((int)positionals.size() >= max_positionals) { errors.push_back("unexpected argument \"" + a + "\""); continue; }
            positionals.push_back(a);
            continue;
        }
        if (a == "--") { options_done = true; continue; }
        if (a.compare(0, 2, "--") == 0) {
            const size_t eq = a.find('=');
            const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            ArgSpec *s = find_long(name);
            if (!s) { errors.push_back("invalid option \"" + a + "\""); continue; }
            if (s->kind == ArgKind::Flag) {
                if (eq != std::string::npos) { errors.push_back("option --" + name + " takes no argument"); continue; }
                if (s->count > 0) { errors.push_back("excess option --" + name); continue; }
                ++s->count;
                continue;
            }
            std::string val;
            if (eq != std::string::npos) { val = a.substr(eq + 1); }
            else if (i + 1 < argc) { val = argv[++i]; }
            else { errors.push_back("option --" + name + " requires an argument"); continue; }
            set_value(*s, val, "--" + name, errors);
            continue;
        }
        // short options, possibly grouped; the first one that takes a value consumes the rest
        for (size_t p = 1; p < a.size(); ++p) {
            ArgSpec *s = find_short(a[p]);
            const std::string shown = std::string("-") + a[p];
            if (!s) { errors.push_back("invalid option \"" + shown + "\""); break; }
            if (s->kind == ArgKind::Flag) {
                if (s->count > 0) { errors.push_back("excess option " + shown); break; }
                ++s->count;
                continue;
            }
            std::string val;
            if (p + 1 < a.size()) { val = a.substr(p + 1); }
            else if (i + 1 < argc) { val = argv[++i]; }
            else { errors.push_back("option " + shown + " requires an argument"); break; }
            set_value(*s, val, shown, errors);
            break;
        }
    }
    return errors.empty();
}

void ArgParser::print_usage(std::ostream &os, const char *positional_help) const
{
    os << "Usage: " << prog;
    for (const auto &s : specs) {
        os << " [-" << s.shortname;
        if (s.kind != ArgKind::Flag) { os << " " << s.datatype; }
        os << "]";
    }
    os << " [<file>] [<file>]\n\n";
    os << "  " << "<file>" << std::string(21, ' ') << positional_help << "\n";
    for (const auto &s : specs) {
        std::string left = std::string("-") + s.shortname + ", --" + s.longname;
        if (s.kind != ArgKind::Flag) { left += std::string("=") + s.datatype; }
        if (left.size() < 25) { left.append(25 - left.size(), ' '); }
        os << "  " << left << " " << s.glossary << "\n";
    }
}

}  // namespace dspcli
