// nc4lite.hpp -- the small part of NetCDF-4 this project needs, written directly on the HDF5 C API.
//
// The reference reads and writes its files through netcdf-cxx4 (src/oct_fileread.cc, src/oct_filewrite.cc); neither
// libnetcdf nor netcdf-cxx4 exists in this image, an HDF5 1.10 library does.  A NetCDF-4 file IS an HDF5 file that
// follows a few conventions, and those are what is implemented here:
//   * dimensions are HDF5 dimension scales (H5DS): a coordinate variable named like its dimension is the scale;
//     a dimension without a variable gets a placeholder scale whose NAME starts with
//     "This is a netCDF dimension but not a netCDF variable."; every scale carries _Netcdf4Dimid;
//   * a variable's dimensions are the scales attached to its axes (DIMENSION_LIST); scalar variables have a scalar
//     dataspace; link and attribute creation order are tracked (the netCDF library requires that to read a file);
//   * text attributes are fixed-length strings, numeric ones scalars or 1-D arrays.
// Reading accepts what GOES-R L1b files contain: chunked + deflate-compressed integer and float variables with
// scale_factor / add_offset style attributes of any numeric type (HDF5 converts on read).
// Host-side I/O only; nothing here touches the GPU.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace nc4lite {

enum class Type { Byte, Short, Int, Float, Double };

class Error : public std::exception {
public:
    explicit Error(std::string m) : msg_(std::move(m)) {}
    const char *what() const noexcept override { return msg_.c_str(); }
private:
    std::string msg_;
};

class Reader {
public:
    explicit Reader(const std::string &path);          // throws Error when the file cannot be opened
    ~Reader();
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;
    bool has_var(const std::string &name) const;
    std::vector<size_t> shape(const std::string &var) const;            // empty for a scalar
    size_t dim_size(const std::string &dim) const;                      // a dimension is the scale dataset of that name
    // whole-variable reads with conversion to the requested memory type
    void read(const std::string &var, short *out) const;
    void read(const std::string &var, int *out) const;
    void read(const std::string &var, float *out) const;
    void read(const std::string &var, double *out) const;
    bool has_att(const std::string &var, const std::string &att) const;
    float att_float(const std::string &var, const std::string &att) const;     // throw Error when missing
    double att_double(const std::string &var, const std::string &att) const;
    int att_int(const std::string &var, const std::string &att) const;
    std::string att_text(const std::string &var, const std::string &att) const;
private:
    int64_t file_ = -1;
};

class Writer {
public:
    explicit Writer(const std::string &path);           // replaces an existing file
    ~Writer();                                           // closes (attaches the dimension scales first)
    Writer(const Writer &) = delete;
    Writer &operator=(const Writer &) = delete;
    void def_dim(const std::string &name, size_t n);
    // dims: names of previously defined dimensions, slowest first; none = scalar.  deflate > 0 chunks by rows and
    // compresses (2-D variables only).
    void def_var(const std::string &name, Type t, const std::vector<std::string> &dims = {}, int deflate = 0);
    void put_att(const std::string &var, const std::string &att, const std::string &text);
    void put_att(const std::string &var, const std::string &att, float v);
    void put_att(const std::string &var, const std::string &att, double v);
    void put_att(const std::string &var, const std::string &att, int v);
    void put_var(const std::string &var, const short *data);
    void put_var(const std::string &var, const int *data);
    void put_var(const std::string &var, const float *data);
    void put_var(const std::string &var, const double *data);
    void close();
private:
    struct Var { int64_t id; std::vector<std::string> dims; };
    int64_t file_ = -1;
    std::vector<std::pair<std::string, size_t>> dims_;
    std::map<std::string, Var> vars_;
    bool closed_ = false;
    int64_t var_id(const std::string &name) const;
};

// One line per top-level variable: name|<i or f><bytes>|<d0>x<d1>...|att=value;att=value;...   (diagnostics / tests)
std::string describe(const std::string &path);

}  // namespace nc4lite
