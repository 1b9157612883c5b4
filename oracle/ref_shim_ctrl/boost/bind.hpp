// TEST INFRASTRUCTURE (oracle/_ref build only).  boost::bind(&Class::method, obj, _1, _2) for the dynamic_reconfigure callback.
#pragma once
#include <functional>
namespace boost {
template <class R, class C, class A1, class A2, class P1, class P2>
std::function<R(A1, A2)> bind(R (C::*m)(A1, A2), C* obj, P1, P2) { return [m, obj](A1 a, A2 b) { return (obj->*m)(a, b); }; }
struct ref_shim_placeholder {};
}  // namespace boost
static const boost::ref_shim_placeholder _1{}, _2{};
