// oracle/shim/boost/shim_all.hpp — the handful of Boost names the reference's headers mention.  TEST INFRASTRUCTURE ONLY
// (see oracle/shim/Eigen/Dense for the rules of this directory).
//   * boost::unordered_set<Key, Hash, Eq, Alloc>: the reference keeps shadow voxels (Eigen::Vector3i) in one
//     (candidate/hand_set.h:84-86) and iterates over it (hand_set.cpp:157-160, 176).  Boost leaves the iteration order
//     unspecified; THIS shim iterates in lexicographic (x, y, z) order — the order oracle/gpd_oracle.cpp fixes as its
//     definition — and ignores Hash / Eq / the bucket count.
//   * boost::hash / hash_combine: only so that hand_set.h:52-63 (a specialisation) compiles.
//   * boost::shared_ptr / make_shared: std::shared_ptr (PCL 1.9's Ptr typedefs, util/plot.h:58).
//   * boost/pool, boost/random: included by hand_set.h:40-47, never used — empty.
#ifndef GPD_REF_SHIM_BOOST
#define GPD_REF_SHIM_BOOST
#include <cstddef>
#include <functional>
#include <memory>
#include <set>

namespace boost {

template <class T>
struct hash {
  std::size_t operator()(const T &v) const { return std::hash<T>()(v); }
};
template <class T>
inline void hash_combine(std::size_t &seed, const T &v) {
  seed ^= boost::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}

namespace shim {
struct LexLess {
  template <class K>
  bool operator()(const K &a, const K &b) const {
    for (int i = 0; i < (int)a.size(); i++)
      if (a(i) != b(i)) return a(i) < b(i);
    return false;
  }
};
}  // namespace shim

template <class Key, class Hash = boost::hash<Key>, class Eq = std::equal_to<Key>, class Alloc = std::allocator<Key>>
class unordered_set {
  typedef std::set<Key, shim::LexLess> Impl;

 public:
  typedef typename Impl::const_iterator const_iterator;
  typedef typename Impl::const_iterator iterator;
  typedef std::size_t size_type;
  unordered_set() {}
  // the reference passes a double here (hand_set.cpp:138: num_shadow_points * 10000)
  explicit unordered_set(size_type /*bucket_count*/) {}
  std::pair<iterator, bool> insert(const Key &k) { return s_.insert(k); }
  const_iterator find(const Key &k) const { return s_.find(k); }
  const_iterator begin() const { return s_.begin(); }
  const_iterator end() const { return s_.end(); }
  size_type size() const { return s_.size(); }
  bool empty() const { return s_.empty(); }
  void clear() { s_.clear(); }

 private:
  Impl s_;
};

template <class T>
using shared_ptr = std::shared_ptr<T>;
using std::make_shared;

}  // namespace boost
#endif
