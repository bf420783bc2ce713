// oracle/shim/gpd_ref_no_jitter.h — force-included (-include) in front of the reference's candidate/hand_set.cpp ONLY.
// TEST INFRASTRUCTURE.  HandSet::shadowVoxelsToPoints (hand_set.cpp:187-204) adds Gaussian jitter drawn from a
// std::mt19937 seeded by std::random_device: irreproducible by construction.  The oracle (and the product) define the
// shadow without jitter; this header renames `normal_distribution` inside that one translation unit to a stand-in that
// always returns 0, so that the reference's own shadow code runs jitter-free.  No other token of the file changes.
#ifndef GPD_REF_NO_JITTER_H
#define GPD_REF_NO_JITTER_H
#include <random>  // first, unrenamed
namespace std {
template <class T>
struct gpd_ref_no_jitter {
  gpd_ref_no_jitter(T, T) {}
  template <class G>
  T operator()(G &) {
    return T(0);
  }
};
}  // namespace std
#define normal_distribution gpd_ref_no_jitter
#endif
