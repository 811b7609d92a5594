// temporary: members not implemented yet
#include "hot_impl.h"
namespace hot {
#define NI(sig) template <class T> sig { throw Error{ HOT_ERR_INVALID, "not implemented yet" }; }
NI(void Ctx<T>::build_hessian())
NI(void Ctx<T>::build_mg())
NI(void Ctx<T>::get_level(int32_t, int32_t*, int32_t*, int32_t*))
NI(void Ctx<T>::get_matrix(int32_t, int32_t*, void*))
NI(void Ctx<T>::get_prolongation(int32_t, int32_t*, void*))
NI(void Ctx<T>::spmv(int32_t, const void*, void*))
NI(void Ctx<T>::restrict_(int32_t, const void*, void*))
NI(void Ctx<T>::prolong(int32_t, const void*, void*))
NI(void Ctx<T>::smooth(int32_t, int32_t, int32_t, double, void*, void*, const void*))
NI(void Ctx<T>::vcycle(const void*, void*))
NI(void Ctx<T>::solve(hot_stats*))
NI(void Ctx<T>::advance(double, hot_stats*))
template struct Ctx<float>;
template struct Ctx<double>;
}
