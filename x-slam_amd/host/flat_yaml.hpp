// flat_yaml.hpp — reader for the flat "key: value" YAML the reference's configs use
// (Experiments/test_xkinect_fusion/configs/ICL_traj2.yaml; read through yaml-cpp's
// config["k"].as<T>() in KinectFusionReconstruction.cpp:12-72 and main.cpp:28-33).  yaml-cpp is
// not a dependency; the files have no nesting, lists or anchors, so this suffices.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>

namespace xs_host {

class FlatYaml {
public:
    static FlatYaml LoadFile(const std::string &path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open config: " + path);
        std::stringstream ss;
        ss << f.rdbuf();
        return Load(ss.str());
    }
    static FlatYaml Load(const std::string &text) {
        FlatYaml y;
        std::istringstream in(text);
        std::string line;
        while (std::getline(in, line)) {
            bool in_quote = false;
            for (size_t i = 0; i < line.size(); ++i) {
                if (line[i] == '"') in_quote = !in_quote;
                if (line[i] == '#' && !in_quote) { line.erase(i); break; }
            }
            const size_t colon = line.find(':');
            if (colon == std::string::npos) continue;
            std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
            if (key.empty()) continue;
            if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'') && val.back() == val.front()) val = val.substr(1, val.size() - 2);
            y.kv_[key] = val;
        }
        return y;
    }
    bool has(const std::string &k) const { return kv_.count(k) != 0; }
    template <class T> T as(const std::string &k) const;
    template <class T> T as(const std::string &k, T dflt) const { return has(k) ? as<T>(k) : dflt; }
    const std::map<std::string, std::string> &items() const { return kv_; }

private:
    const std::string &raw(const std::string &k) const {
        auto it = kv_.find(k);
        if (it == kv_.end()) throw std::runtime_error("missing config key: " + k);
        return it->second;
    }
    static std::string trim(const std::string &s) {
        const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
    std::map<std::string, std::string> kv_;
};
template <> inline int FlatYaml::as<int>(const std::string &k) const { return (int)std::strtol(raw(k).c_str(), nullptr, 10); }
template <> inline float FlatYaml::as<float>(const std::string &k) const { return std::strtof(raw(k).c_str(), nullptr); }
template <> inline std::string FlatYaml::as<std::string>(const std::string &k) const { return raw(k); }
template <> inline bool FlatYaml::as<bool>(const std::string &k) const {
    const std::string &v = raw(k);
    return v == "true" || v == "True" || v == "TRUE" || v == "yes" || v == "on" || v == "1";
}

}  // namespace xs_host
