// TEST INFRASTRUCTURE: VisionTools' checked container access as the reference uses it (maths_utils.cpp includes the header and uses nothing of it;
// matcher.cpp, pose_optimizer.h and slam_graph.cpp look elements up with it).  GET_MAP_ELEM(key, map) = the element that must exist: a value
// by reference, a raw pointer as it is, the POINTEE of a shared pointer (slam_graph.cpp binds `const Vertex &` to an element of a table of
// VertexPtr); GET_MAP_ELEM_REF(key, pointer to map) = the same, writable; GET_VEC_VAL_REF(index, pointer to vector).
#pragma once
#include <cassert>
#include <tr1/memory>
#include <utility>
namespace VisionTools {
template <class T> const T &svs_elem(const T &v) { return v; }
template <class T> T *svs_elem(T *const &p) { return p; }
template <class T> const T &svs_elem(const std::tr1::shared_ptr<T> &p) { return *p; }
template <class T> T &svs_elem_ref(T &v) { return v; }
template <class T> T &svs_elem_ref(std::tr1::shared_ptr<T> &p) { return *p; }
template <class K, class M> decltype(auto) svs_get_map_elem(const K &k, const M &m) {
  typename M::const_iterator it = m.find(k);
  assert(it != m.end());
  return svs_elem(it->second);
}
template <class K, class M> decltype(auto) svs_get_map_elem_ref(const K &k, M *m) {
  typename M::iterator it = m->find(k);
  assert(it != m->end());
  return svs_elem_ref(it->second);
}
}
#define GET_MAP_ELEM(key, map) VisionTools::svs_get_map_elem(key, map)
#define GET_MAP_ELEM_REF(key, map) VisionTools::svs_get_map_elem_ref(key, map)
#define GET_VEC_VAL_REF(idx, vec) ((vec)->at(idx))
// IS_IN_SET(key, container): membership by find(); ADD_TO_MAP_ELEM(key, val, pointer to map): map[key] += val, the element made (from val) when absent --
// as StereoFrontend::addNewKeyframe / shallWeSwitchKeyframe (stereo_frontend.cpp:309-510) use them
namespace VisionTools {
template <class K, class M, class V> void svs_add_to_map_elem(const K &k, const V &v, M *m) {
  typename M::iterator it = m->find(k);
  if (it == m->end()) m->insert(std::make_pair(k, v));
  else it->second += v;
}
}
#define IS_IN_SET(key, set) ((set).find(key) != (set).end())
#define ADD_TO_MAP_ELEM(key, val, map) VisionTools::svs_add_to_map_elem(key, val, map)
