// TEST INFRASTRUCTURE: stand-in for pangolin::Var<T> (a handle to a named, live GUI variable): reads as the default it was constructed with
// unless the wrapper has put a value for that name into svs_shim_var_override -- looked up at every read, as the slider is live (the
// front-end keeps its handles in function-local statics)
#pragma once
#include <map>
#include <string>
extern std::map<std::string, double> svs_shim_var_override;
namespace pangolin {
template <typename T> struct Var {
  std::string name; T def;
  Var(const std::string &name_, T def_ = T(), T lo = T(), T hi = T()) : name(name_), def(def_) {}
  operator T() const {
    std::map<std::string, double>::const_iterator it = svs_shim_var_override.find(name);
    return it != svs_shim_var_override.end() ? (T)it->second : def;
  }
};
}
