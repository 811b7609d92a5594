// libhotmi355x — frame output of the particle state (host code).
//
//   hot_write_partio   MpmSimulationBase::writeState -> writePartio (Lib/MPM/MpmSimulationBase.cpp:754-764,
//                      Lib/Ziran/Math/Geometry/PartioIO.h:142-180): a Houdini .bgeo ("Bgeo" v5, big-endian) holding the particle
//                      positions as floats — the only attribute the reference writes.  The container is written here directly
//                      (partio is not available): header, no extra point attributes, one (x, y, z, 1) float quadruple per point,
//                      the two-byte extra block.
//   hot_write_restart / hot_read_restart
//                      SimulationBase::write / read -> Scene::writeState -> DataManager::writeData (Lib/Ziran/Sim/SimulationBase.h:
//                      152-190, Lib/Ziran/Sim/Scene.h:189-219, Lib/Ziran/CS/DataStructure/DataManager.h:263-294, DataArray.h:100-105,
//                      Lib/Ziran/CS/Util/BinaryIO.h:82-88): the DataManager container — count, number of arrays, then per array its
//                      name, lg2_grain_size, the DisjointRanges vector and the value vector (size, sizeof, raw little-endian
//                      values) — with the columns of this library: "m", "P", "V", "C", "F", "element measure" under the
//                      reference's names and "mu", "lambda", "Jp" for the per-particle fixed-corotated parameters.  The reference
//                      serialises its constitutive-model objects and element managers instead of those three columns, so the two
//                      restart files share the container layout but are not interchangeable.
// Particles are written in the caller's original order (particle_order undone).  A sharded context writes its own shard.
#include "hot_impl.h"
#include <fstream>

namespace hot {

template <class U>
static void put(std::ostream& o, const U& v)
{
    o.write(reinterpret_cast<const char*>(&v), sizeof(U));
}
template <class U>
static U get(std::istream& in)
{
    U v{};
    in.read(reinterpret_cast<char*>(&v), sizeof(U));
    return v;
}
static void put_be32(std::ostream& o, uint32_t v)
{
    unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v };
    o.write(reinterpret_cast<const char*>(b), 4);
}
static void put_be_float(std::ostream& o, float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    put_be32(o, u);
}

template <class T>
void Ctx<T>::write_partio(const char* path)
{
    need(Np > 0 && path, "hot_write_partio: no particles / no path");
    std::vector<T> X(3 * (size_t)Np);
    get_particles(X.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    std::ofstream o(path, std::ios::binary);
    HOT_CHECK(o.good(), HOT_ERR_INVALID, std::string("hot_write_partio: cannot open ") + path);
    put_be32(o, (((('B' << 8) | 'g') << 8 | 'e') << 8) | 'o'); // magic
    o.put('V');
    put_be32(o, 5); // version
    put_be32(o, (uint32_t)Np); // nPoints
    for (int k = 0; k < 3; ++k) put_be32(o, 0); // nPrims, nPointGroups, nPrimGroups
    for (int k = 0; k < 4; ++k) put_be32(o, 0); // nPointAttrib (besides position), nVertexAttrib, nPrimAttrib, nAttrib
    for (int64_t p = 0; p < Np; ++p) {
        for (int d = 0; d < 3; ++d) put_be_float(o, (float)X[3 * p + d]);
        put_be_float(o, 1.0f); // homogeneous coordinate
    }
    o.put((char)0x00), o.put((char)0xff); // beginExtra, endExtra
    HOT_CHECK(o.good(), HOT_ERR_INVALID, "hot_write_partio: write failed");
}

static void put_string(std::ostream& o, const std::string& s)
{
    put<uint64_t>(o, s.size());
    o.write(s.data(), (std::streamsize)s.size());
}
template <class T>
static void put_array(std::ostream& o, const std::string& name, const std::vector<T>& v, int comps, int64_t count)
{
    put_string(o, name);
    put<int32_t>(o, 7); // lg2_grain_size
    put<uint64_t>(o, 1), put<uint64_t>(o, 8); // one Range {lower, upper}
    put<int32_t>(o, 0), put<int32_t>(o, (int32_t)count);
    put<uint64_t>(o, (uint64_t)count), put<uint64_t>(o, (uint64_t)comps * sizeof(T));
    o.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

template <class T>
void Ctx<T>::write_restart(const char* path)
{
    need(Np > 0 && path, "hot_write_restart: no particles / no path");
    const size_t n = (size_t)Np;
    std::vector<T> X(3 * n), V(3 * n), C(9 * n), F(9 * n), mu(n), la(n), jp(n), m(n), vol(n);
    get_particles(X.data(), V.data(), C.data(), F.data(), mu.data(), la.data(), jp.data());
    // mass and volume are not part of hot_get_particles: fetch them in original order through the same permutation
    {
        std::vector<T> hm(n), hv(n);
        std::vector<int32_t> s2o(n);
        HOT_HIP(hipMemcpyAsync(hm.data(), pM.p, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        HOT_HIP(hipMemcpyAsync(hv.data(), pVol.p, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        HOT_HIP(hipMemcpyAsync(s2o.data(), slot2orig.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        sync();
        for (size_t p = 0; p < n; ++p) m[s2o[p]] = hm[p], vol[s2o[p]] = hv[p];
    }
    std::ofstream o(path, std::ios::binary);
    HOT_CHECK(o.good(), HOT_ERR_INVALID, std::string("hot_write_restart: cannot open ") + path);
    put<int32_t>(o, (int32_t)Np); // DataManager::count
    put<uint64_t>(o, 9); // number of arrays
    put_array(o, "m", m, 1, Np), put_array(o, "P", X, 3, Np), put_array(o, "V", V, 3, Np), put_array(o, "C", C, 9, Np), put_array(o, "F", F, 9, Np);
    put_array(o, "element measure", vol, 1, Np), put_array(o, "mu", mu, 1, Np), put_array(o, "lambda", la, 1, Np), put_array(o, "Jp", jp, 1, Np);
    HOT_CHECK(o.good(), HOT_ERR_INVALID, "hot_write_restart: write failed");
}

template <class T>
void Ctx<T>::read_restart(const char* path)
{
    need(path, "hot_read_restart: no path");
    std::ifstream in(path, std::ios::binary);
    HOT_CHECK(in.good(), HOT_ERR_INVALID, std::string("hot_read_restart: cannot open ") + path);
    const int64_t count = get<int32_t>(in);
    const uint64_t narr = get<uint64_t>(in);
    HOT_CHECK(count > 0 && narr < 64, HOT_ERR_INVALID, "hot_read_restart: not a restart file of this library");
    std::map<std::string, std::vector<T>> col;
    for (uint64_t a = 0; a < narr; ++a) {
        const uint64_t len = get<uint64_t>(in);
        HOT_CHECK(len < 256, HOT_ERR_INVALID, "hot_read_restart: corrupt array name");
        std::string name(len, ' ');
        in.read(&name[0], (std::streamsize)len);
        (void)get<int32_t>(in); // lg2_grain_size
        const uint64_t nr = get<uint64_t>(in), rb = get<uint64_t>(in);
        in.seekg((std::streamoff)(nr * rb), std::ios::cur); // the ranges: one contiguous range is all this library writes
        const uint64_t cnt = get<uint64_t>(in), bytes = get<uint64_t>(in);
        HOT_CHECK((int64_t)cnt == count && bytes % sizeof(T) == 0 && bytes <= 9 * sizeof(T), HOT_ERR_INVALID, "hot_read_restart: array '" + name + "' has the wrong length or scalar type (the file was written with the other precision?)");
        std::vector<T>& v = col[name];
        v.resize(cnt * (bytes / sizeof(T)));
        in.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
        HOT_CHECK(in.good(), HOT_ERR_INVALID, "hot_read_restart: truncated file");
    }
    const std::pair<const char*, int> want[] = { { "m", 1 }, { "P", 3 }, { "V", 3 }, { "C", 9 }, { "F", 9 }, { "element measure", 1 }, { "mu", 1 }, { "lambda", 1 }, { "Jp", 1 } };
    for (const auto& kw : want) {
        HOT_CHECK(col.count(kw.first) == 1, HOT_ERR_INVALID, std::string("hot_read_restart: array missing: ") + kw.first);
        HOT_CHECK((int64_t)col[kw.first].size() == count * kw.second, HOT_ERR_INVALID, std::string("hot_read_restart: array has the wrong width: ") + kw.first);
    }
    set_particles(count, col["P"].data(), col["V"].data(), col["m"].data(), col["C"].data(), col["F"].data(), col["element measure"].data(), col["mu"].data(), col["lambda"].data(), col["Jp"].data());
}

template struct Ctx<float>;
template struct Ctx<double>;

} // namespace hot
