// util::ConfigFile — flat `key = value` parser, `#` comments (reference: util/config_file.cpp:76-167,
// config_file.h:82-143).  Same method names and defaults-on-missing behaviour.
#pragma once
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace gpd {
namespace util {

class ConfigFile {
 public:
  explicit ConfigFile(const std::string &fName) : fName_(fName) {}
  bool ExtractKeys();
  bool keyExists(const std::string &key) const { return contents_.find(key) != contents_.end(); }
  template <typename T>
  T getValueOfKey(const std::string &key, T const &defaultValue) const {
    auto it = contents_.find(key);
    if (it == contents_.end()) return defaultValue;
    std::istringstream is(it->second);
    T v;
    if (!(is >> v)) return defaultValue;
    return v;
  }
  std::string getValueOfKeyAsString(const std::string &key, const std::string &defaultValue) const;
  std::vector<double> getValueOfKeyAsStdVectorDouble(const std::string &key, const std::string &defaultValue) const;
  std::vector<int> getValueOfKeyAsStdVectorInt(const std::string &key, const std::string &defaultValue) const;

 private:
  std::string fName_;
  std::map<std::string, std::string> contents_;
};

}  // namespace util
}  // namespace gpd
