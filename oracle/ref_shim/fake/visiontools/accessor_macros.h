// TEST INFRASTRUCTURE: VisionTools' checked map access GET_MAP_ELEM(key, map) = the element that must exist (maths_utils.cpp includes the
// header and uses nothing of it; matcher.cpp looks keyframes and vertices up with it)
#pragma once
#include <cassert>
namespace VisionTools {
template <class K, class M> const typename M::mapped_type &svs_get_map_elem(const K &k, const M &m) {
  typename M::const_iterator it = m.find(k);
  assert(it != m.end());
  return it->second;
}
}
#define GET_MAP_ELEM(key, map) VisionTools::svs_get_map_elem(key, map)
// GET_VEC_VAL_REF(index, pointer to vector) = reference to the element that must exist (pose_optimizer.h:159)
#define GET_VEC_VAL_REF(idx, vec) ((vec)->at(idx))
