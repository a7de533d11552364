// str_fun.h -- the two text helpers of the scene.txt format (str_fun.h:11-40 in the reference).
#ifndef PT_HOST_STR_FUN_H
#define PT_HOST_STR_FUN_H
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

// "key = value" -> {key, value} with every space removed.  A line without '=' yields the whole
// (space-stripped) line twice, which is what the reference's find()/substr() arithmetic produces.
inline std::vector<std::string> get_key_value(std::string line)
{
    line.erase(std::remove(line.begin(), line.end(), ' '), line.end());
    if (!line.empty() && line.back() == '\r') line.pop_back(); // files written on Windows
    std::vector<std::string> kv;
    auto eq = line.find('=');
    if (eq == std::string::npos) {
        kv.push_back(line);
        kv.push_back(line);
    } else {
        kv.push_back(line.substr(0, eq));
        kv.push_back(line.substr(eq + 1));
    }
    return kv;
}

// Splits at every `sep`, dropping empty pieces.
inline std::vector<std::string> explode(const std::string &s, const char &sep)
{
    std::vector<std::string> out;
    std::string cur;
    for (char ch : s) {
        if (ch != sep) cur += ch;
        else if (!cur.empty()) { out.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}
#endif
