// nc4lite.cpp -- see nc4lite.hpp.
#include "nc4lite.hpp"

#include <hdf5.h>
#include <hdf5_hl.h>

#include <cstdio>
#include <cstring>

namespace nc4lite {

namespace {

struct Quiet {          // HDF5 prints its error stack by default; errors are reported through exceptions here
    Quiet() { H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); }
};
static Quiet g_quiet;

hid_t file_type(Type t)
{
    switch (t) {
    case Type::Byte: return H5T_STD_I8LE;
    case Type::Short: return H5T_STD_I16LE;
    case Type::Int: return H5T_STD_I32LE;
    case Type::Float: return H5T_IEEE_F32LE;
    default: return H5T_IEEE_F64LE;
    }
}

struct Hid {            // close-on-scope-exit for the handle kinds used below
    hid_t id;
    int kind;           // 0 dataset, 1 dataspace, 2 attribute, 3 type, 4 property list
    Hid(hid_t i, int k) : id(i), kind(k) {}
    ~Hid()
    {
        if (id < 0) return;
        switch (kind) {
        case 0: H5Dclose(id); break;
        case 1: H5Sclose(id); break;
        case 2: H5Aclose(id); break;
        case 3: H5Tclose(id); break;
        default: H5Pclose(id); break;
        }
    }
    Hid(const Hid &) = delete;
    Hid &operator=(const Hid &) = delete;
};

void read_all(hid_t file, const std::string &var, hid_t mtype, void *out)
{
    Hid d(H5Dopen2(file, var.c_str(), H5P_DEFAULT), 0);
    if (d.id < 0) throw Error("variable '" + var + "' not found");
    if (H5Dread(d.id, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0) throw Error("reading variable '" + var + "' failed");
}

void read_att(hid_t file, const std::string &var, const std::string &att, hid_t mtype, void *out)
{
    Hid d(H5Dopen2(file, var.c_str(), H5P_DEFAULT), 0);
    if (d.id < 0) throw Error("variable '" + var + "' not found");
    Hid a(H5Aopen(d.id, att.c_str(), H5P_DEFAULT), 2);
    if (a.id < 0) throw Error("attribute '" + att + "' of '" + var + "' not found");
    Hid sp(H5Aget_space(a.id), 1);
    if (H5Sget_simple_extent_npoints(sp.id) < 1) throw Error("attribute '" + att + "' of '" + var + "' is empty");
    // the first value of an array-valued attribute, as netcdf-cxx4's getValues(&scalar) does
    const hssize_t n = H5Sget_simple_extent_npoints(sp.id);
    std::vector<double> buf((size_t)n);          // large enough for any numeric type
    if (H5Aread(a.id, mtype, buf.data()) < 0) throw Error("reading attribute '" + att + "' of '" + var + "' failed");
    std::memcpy(out, buf.data(), H5Tget_size(mtype));
}

}  // namespace

// --------------------------------------------------------------------------------------------- Reader
Reader::Reader(const std::string &path)
{
    file_ = H5Fopen(path.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file_ < 0) throw Error("cannot open '" + path + "' (not there, or not a NetCDF-4 / HDF5 file)");
}
Reader::~Reader() { if (file_ >= 0) H5Fclose((hid_t)file_); }

bool Reader::has_var(const std::string &name) const
{
    return H5Lexists((hid_t)file_, name.c_str(), H5P_DEFAULT) > 0;
}

std::vector<size_t> Reader::shape(const std::string &var) const
{
    Hid d(H5Dopen2((hid_t)file_, var.c_str(), H5P_DEFAULT), 0);
    if (d.id < 0) throw Error("variable '" + var + "' not found");
    Hid sp(H5Dget_space(d.id), 1);
    const int nd = H5Sget_simple_extent_ndims(sp.id);
    std::vector<hsize_t> dims(nd > 0 ? nd : 0);
    if (nd > 0) H5Sget_simple_extent_dims(sp.id, dims.data(), nullptr);
    return std::vector<size_t>(dims.begin(), dims.end());
}

size_t Reader::dim_size(const std::string &dim) const
{
    if (!has_var(dim)) throw Error("dimension '" + dim + "' not found");
    const std::vector<size_t> s = shape(dim);
    if (s.size() != 1) throw Error("'" + dim + "' is not a dimension");
    return s[0];
}

void Reader::read(const std::string &var, short *out) const { read_all((hid_t)file_, var, H5T_NATIVE_SHORT, out); }
void Reader::read(const std::string &var, int *out) const { read_all((hid_t)file_, var, H5T_NATIVE_INT, out); }
void Reader::read(const std::string &var, float *out) const { read_all((hid_t)file_, var, H5T_NATIVE_FLOAT, out); }
void Reader::read(const std::string &var, double *out) const { read_all((hid_t)file_, var, H5T_NATIVE_DOUBLE, out); }

bool Reader::has_att(const std::string &var, const std::string &att) const
{
    Hid d(H5Dopen2((hid_t)file_, var.c_str(), H5P_DEFAULT), 0);
    if (d.id < 0) return false;
    return H5Aexists(d.id, att.c_str()) > 0;
}

float Reader::att_float(const std::string &var, const std::string &att) const
{
    float v = 0.f;
    read_att((hid_t)file_, var, att, H5T_NATIVE_FLOAT, &v);
    return v;
}
double Reader::att_double(const std::string &var, const std::string &att) const
{
    double v = 0.;
    read_att((hid_t)file_, var, att, H5T_NATIVE_DOUBLE, &v);
    return v;
}
int Reader::att_int(const std::string &var, const std::string &att) const
{
    int v = 0;
    read_att((hid_t)file_, var, att, H5T_NATIVE_INT, &v);
    return v;
}

std::string Reader::att_text(const std::string &var, const std::string &att) const
{
    Hid d(H5Dopen2((hid_t)file_, var.c_str(), H5P_DEFAULT), 0);
    if (d.id < 0) throw Error("variable '" + var + "' not found");
    Hid a(H5Aopen(d.id, att.c_str(), H5P_DEFAULT), 2);
    if (a.id < 0) throw Error("attribute '" + att + "' of '" + var + "' not found");
    Hid t(H5Aget_type(a.id), 3);
    if (H5Tget_class(t.id) != H5T_STRING) throw Error("attribute '" + att + "' of '" + var + "' is not text");
    std::string out;
    if (H5Tis_variable_str(t.id) > 0) {
        char *p = nullptr;
        Hid mt(H5Tcopy(H5T_C_S1), 3);
        H5Tset_size(mt.id, H5T_VARIABLE);
        H5Tset_cset(mt.id, H5Tget_cset(t.id));
        if (H5Aread(a.id, mt.id, &p) < 0) throw Error("reading attribute '" + att + "' failed");
        if (p) { out = p; H5free_memory(p); }
    } else {
        const size_t n = H5Tget_size(t.id);
        std::vector<char> buf(n + 1, 0);
        Hid mt(H5Tcopy(H5T_C_S1), 3);
        H5Tset_size(mt.id, n);
        H5Tset_cset(mt.id, H5Tget_cset(t.id));
        H5Tset_strpad(mt.id, H5Tget_strpad(t.id));
        if (H5Aread(a.id, mt.id, buf.data()) < 0) throw Error("reading attribute '" + att + "' failed");
        out.assign(buf.data(), strnlen(buf.data(), n));
    }
    return out;
}

// --------------------------------------------------------------------------------------------- Writer
Writer::Writer(const std::string &path)
{
    Hid fcpl(H5Pcreate(H5P_FILE_CREATE), 4);
    H5Pset_link_creation_order(fcpl.id, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    H5Pset_attr_creation_order(fcpl.id, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    file_ = H5Fcreate(path.c_str(), H5F_ACC_TRUNC, fcpl.id, H5P_DEFAULT);
    if (file_ < 0) throw Error("cannot create '" + path + "'");
}

Writer::~Writer()
{
    try { close(); } catch (...) {}
}

void Writer::def_dim(const std::string &name, size_t n)
{
    for (auto &d : dims_)
        if (d.first == name) throw Error("dimension '" + name + "' defined twice");
    dims_.emplace_back(name, n);
}

int64_t Writer::var_id(const std::string &name) const
{
    auto it = vars_.find(name);
    if (it == vars_.end()) throw Error("variable '" + name + "' not defined");
    return it->second.id;
}

void Writer::def_var(const std::string &name, Type t, const std::vector<std::string> &dims, int deflate)
{
    if (vars_.count(name)) throw Error("variable '" + name + "' defined twice");
    std::vector<hsize_t> ext;
    for (auto &dn : dims) {
        bool found = false;
        for (auto &d : dims_)
            if (d.first == dn) { ext.push_back(d.second); found = true; }
        if (!found) throw Error("dimension '" + dn + "' of variable '" + name + "' not defined");
    }
    Hid sp(ext.empty() ? H5Screate(H5S_SCALAR) : H5Screate_simple((int)ext.size(), ext.data(), nullptr), 1);
    Hid dcpl(H5Pcreate(H5P_DATASET_CREATE), 4);
    H5Pset_attr_creation_order(dcpl.id, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    if (deflate > 0 && ext.size() == 2) {
        hsize_t chunk[2] = {ext[0] < 256 ? ext[0] : 256, ext[1]};
        H5Pset_chunk(dcpl.id, 2, chunk);
        H5Pset_shuffle(dcpl.id);
        H5Pset_deflate(dcpl.id, (unsigned)deflate);
    }
    const hid_t d = H5Dcreate2((hid_t)file_, name.c_str(), file_type(t), sp.id, H5P_DEFAULT, dcpl.id, H5P_DEFAULT);
    if (d < 0) throw Error("cannot create variable '" + name + "'");
    vars_[name] = Var{d, dims};
}

void Writer::put_att(const std::string &var, const std::string &att, const std::string &text)
{
    const hid_t d = (hid_t)var_id(var);
    if (H5Aexists(d, att.c_str()) > 0) H5Adelete(d, att.c_str());      // netcdf-cxx4's putAtt overwrites
    Hid t(H5Tcopy(H5T_C_S1), 3);
    H5Tset_size(t.id, text.empty() ? 1 : text.size());
    H5Tset_strpad(t.id, H5T_STR_NULLTERM);
    Hid sp(H5Screate(H5S_SCALAR), 1);
    Hid a(H5Acreate2(d, att.c_str(), t.id, sp.id, H5P_DEFAULT, H5P_DEFAULT), 2);
    if (a.id < 0 || H5Awrite(a.id, t.id, text.empty() ? "" : text.c_str()) < 0)
        throw Error("writing attribute '" + att + "' of '" + var + "' failed");
}

namespace {
void put_num_att(hid_t d, const std::string &var, const std::string &att, hid_t ftype, hid_t mtype, const void *v)
{
    if (H5Aexists(d, att.c_str()) > 0) H5Adelete(d, att.c_str());
    hsize_t one = 1;
    Hid sp(H5Screate_simple(1, &one, nullptr), 1);     // netCDF stores numeric attributes as 1-D arrays
    Hid a(H5Acreate2(d, att.c_str(), ftype, sp.id, H5P_DEFAULT, H5P_DEFAULT), 2);
    if (a.id < 0 || H5Awrite(a.id, mtype, v) < 0) throw Error("writing attribute '" + att + "' of '" + var + "' failed");
}
}  // namespace

void Writer::put_att(const std::string &var, const std::string &att, float v)
{
    put_num_att((hid_t)var_id(var), var, att, H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, &v);
}
void Writer::put_att(const std::string &var, const std::string &att, double v)
{
    put_num_att((hid_t)var_id(var), var, att, H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, &v);
}
void Writer::put_att(const std::string &var, const std::string &att, int v)
{
    put_num_att((hid_t)var_id(var), var, att, H5T_STD_I32LE, H5T_NATIVE_INT, &v);
}

void Writer::put_var(const std::string &var, const short *data)
{
    if (H5Dwrite((hid_t)var_id(var), H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw Error("writing '" + var + "' failed");
}
void Writer::put_var(const std::string &var, const int *data)
{
    if (H5Dwrite((hid_t)var_id(var), H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw Error("writing '" + var + "' failed");
}
void Writer::put_var(const std::string &var, const float *data)
{
    if (H5Dwrite((hid_t)var_id(var), H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw Error("writing '" + var + "' failed");
}
void Writer::put_var(const std::string &var, const double *data)
{
    if (H5Dwrite((hid_t)var_id(var), H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw Error("writing '" + var + "' failed");
}

void Writer::close()
{
    if (closed_) return;
    closed_ = true;
    // dimensions become dimension scales; a dimension without a coordinate variable gets a placeholder dataset
    std::map<std::string, hid_t> scale;
    int dimid = 0;
    for (auto &d : dims_) {
        hid_t s;
        auto it = vars_.find(d.first);
        if (it != vars_.end()) {
            s = (hid_t)it->second.id;
            H5DSset_scale(s, d.first.c_str());
        } else {
            hsize_t n = d.second;
            Hid sp(H5Screate_simple(1, &n, nullptr), 1);
            Hid dcpl(H5Pcreate(H5P_DATASET_CREATE), 4);
            H5Pset_attr_creation_order(dcpl.id, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
            s = H5Dcreate2((hid_t)file_, d.first.c_str(), H5T_IEEE_F32BE, sp.id, H5P_DEFAULT, dcpl.id, H5P_DEFAULT);
            char nm[96];
            std::snprintf(nm, sizeof nm, "This is a netCDF dimension but not a netCDF variable.%10d", (int)d.second);
            H5DSset_scale(s, nm);
            vars_[d.first] = Var{s, {}};
        }
        {
            Hid sp(H5Screate(H5S_SCALAR), 1);
            Hid a(H5Acreate2(s, "_Netcdf4Dimid", H5T_STD_I32LE, sp.id, H5P_DEFAULT, H5P_DEFAULT), 2);
            if (a.id >= 0) H5Awrite(a.id, H5T_NATIVE_INT, &dimid);
        }
        scale[d.first] = s;
        dimid++;
    }
    for (auto &kv : vars_) {
        const Var &v = kv.second;
        for (size_t ax = 0; ax < v.dims.size(); ax++) {
            auto it = scale.find(v.dims[ax]);
            if (it == scale.end() || it->second == (hid_t)v.id) continue;       // a coordinate variable is its own scale
            H5DSattach_scale((hid_t)v.id, it->second, (unsigned)ax);
        }
    }
    for (auto &kv : vars_) H5Dclose((hid_t)kv.second.id);
    vars_.clear();
    if (file_ >= 0) { H5Fclose((hid_t)file_); file_ = -1; }
}

namespace {

herr_t describe_att(hid_t loc, const char *name, const H5A_info_t *, void *out_)
{
    std::string &out = *static_cast<std::string *>(out_);
    Hid a(H5Aopen(loc, name, H5P_DEFAULT), 2);
    Hid t(H5Aget_type(a.id), 3);
    char buf[64];
    if (H5Tget_class(t.id) == H5T_STRING) {
        if (H5Tis_variable_str(t.id) > 0) { out += std::string(name) + "=<vlen>;"; return 0; }
        const size_t n = H5Tget_size(t.id);
        std::vector<char> b(n + 1, 0);
        H5Aread(a.id, t.id, b.data());
        out += std::string(name) + "=" + b.data() + ";";
    } else if (H5Tget_class(t.id) == H5T_INTEGER || H5Tget_class(t.id) == H5T_FLOAT) {
        Hid sp(H5Aget_space(a.id), 1);
        const hssize_t n = H5Sget_simple_extent_npoints(sp.id);
        std::vector<double> v((size_t)(n > 0 ? n : 1));
        H5Aread(a.id, H5T_NATIVE_DOUBLE, v.data());
        std::snprintf(buf, sizeof buf, "%.9g", v[0]);
        out += std::string(name) + "=" + buf + ";";
    } else {
        out += std::string(name) + "=<other>;";
    }
    return 0;
}

herr_t describe_var(hid_t loc, const char *name, const H5L_info_t *, void *out_)
{
    std::string &out = *static_cast<std::string *>(out_);
    Hid d(H5Dopen2(loc, name, H5P_DEFAULT), 0);
    if (d.id < 0) return 0;
    Hid sp(H5Dget_space(d.id), 1);
    const int nd = H5Sget_simple_extent_ndims(sp.id);
    hsize_t dims[8] = {0};
    if (nd > 0 && nd <= 8) H5Sget_simple_extent_dims(sp.id, dims, nullptr);
    Hid t(H5Dget_type(d.id), 3);
    char buf[64];
    std::snprintf(buf, sizeof buf, "|%s%zu|", H5Tget_class(t.id) == H5T_FLOAT ? "f" : "i", H5Tget_size(t.id));
    out += std::string(name) + buf;
    for (int i = 0; i < nd && i < 8; i++) { std::snprintf(buf, sizeof buf, "%s%llu", i ? "x" : "", (unsigned long long)dims[i]); out += buf; }
    out += "|";
    hsize_t idx = 0;
    H5Aiterate2(d.id, H5_INDEX_NAME, H5_ITER_NATIVE, &idx, describe_att, &out);
    out += "\n";
    return 0;
}

}  // namespace

std::string describe(const std::string &path)
{
    const hid_t f = H5Fopen(path.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) throw Error("cannot open '" + path + "'");
    std::string out;
    hsize_t idx = 0;
    H5Literate(f, H5_INDEX_NAME, H5_ITER_NATIVE, &idx, describe_var, &out);
    H5Fclose(f);
    return out;
}

}  // namespace nc4lite
