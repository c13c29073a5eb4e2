// mm_hostutil.h — small host-side string helpers shared by the cold-path sources.
#ifndef MM_HOSTUTIL_H
#define MM_HOSTUTIL_H
#include <cctype>
#include <cstdint>
#include <sstream>
#include <string>
#include <vector>

namespace mmh {

inline void ltrim(std::string &s) {
  size_t i = 0;
  while (i < s.size() && std::isspace((unsigned char) s[i])) ++i;
  s.erase(0, i);
}
inline void rtrim(std::string &s) {
  size_t n = s.size();
  while (n > 0 && std::isspace((unsigned char) s[n - 1])) --n;
  s.erase(n);
}
inline void trim(std::string &s) { ltrim(s); rtrim(s); }

// split at the first occurrence of `delim`; both halves trimmed; false if `delim` is absent
inline bool split_first(const std::string &in, char delim, std::string &a, std::string &b) {
  const size_t p = in.find(delim);
  if (p == std::string::npos) return false;
  a = in.substr(0, p); b = in.substr(p + 1);
  trim(a); trim(b);
  return true;
}
// getline-style split: a trailing empty field is not produced, an empty input gives no field
inline void split_all(const std::string &in, char delim, std::vector<std::string> &out) {
  out.clear();
  std::stringstream ss(in);
  std::string item;
  while (std::getline(ss, item, delim)) out.push_back(item);
}
// number with ',' thousands separators (the reference imbues cerr with such a locale, mmannot.cpp:111-115, 2092)
inline std::string commas(uint64_t v) {
  std::string d = std::to_string(v), o;
  for (size_t i = 0; i < d.size(); ++i) {
    if (i && (d.size() - i) % 3 == 0) o += ',';
    o += d[i];
  }
  return o;
}

}  // namespace mmh
#endif
