/*
 * Splat input from PLY files: FastPly::Reader (src/fast_ply.h:77-262, src/fast_ply.cpp:66-350) -- row f4 of
 * SURVEY.md section 8, host side only (no device code in this file).
 *
 * Accepts what the reference accepts: binary PLY in the CPU's byte order, version 1.0, `vertex` as the first
 * element with float32 properties x y z nx ny nz radius in any order and at any offset among other scalar
 * properties; lists in the vertex element are rejected.  A splat's radius is min(radius, maxRadius) * smooth and
 * its quality 1 / radius^2 (Reader::decode, :334-350).  Errors carry the reference's FormatError texts.
 */
#include "common.hpp"
#include "placement.hpp"

#include <cstdio>
#include <algorithm>
#include <fstream>
#include <iterator>
#include <memory>
#include <sstream>
#include <thread>

using namespace mlsgpu;

namespace
{

enum FieldType { INT8, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64, BAD_TYPE };

/* the scalar types of the format: spelling -> (type, bytes, may count a list) */
struct TypeRow
{
    const char *spelling;
    FieldType type;
    uint64_t bytes;
    bool integral;
};
const TypeRow typeRows[] = {
    {"int8", INT8, 1, true},       {"char", INT8, 1, true},     {"uint8", UINT8, 1, true},    {"uchar", UINT8, 1, true},
    {"int16", INT16, 2, true},     {"uint16", UINT16, 2, true}, {"int32", INT32, 4, true},    {"int", INT32, 4, true},
    {"uint32", UINT32, 4, true},   {"uint", UINT32, 4, true},   {"float32", FLOAT32, 4, false}, {"float", FLOAT32, 4, false},
    {"float64", FLOAT64, 8, false},
};

const TypeRow *findType(const std::string &spelling)
{
    for (const TypeRow &row : typeRows)
        if (spelling == row.spelling)
            return &row;
    return nullptr;
}

enum { X, Y, Z, NX, NY, NZ, RADIUS, NUM_PROPERTIES };

} // namespace

struct mlsgpu_ply_reader
{
    std::string path;
    std::ifstream in;
    float smooth = 1.0f, maxRadius = 0.0f;
    uint64_t vertexSize = 0, vertexCount = 0, headerSize = 0;
    uint64_t offsets[NUM_PROPERTIES] = {0, 0, 0, 0, 0, 0, 0};
    std::vector<char> buffer;
};

namespace
{

int formatError(const mlsgpu_ply_reader &r, const std::string &what)
{
    return setError(MLSGPU_ERR_FORMAT, "%s: %s", r.path.c_str(), what.c_str());
}

/*
 * The header (what Reader::readHeader accepts, src/fast_ply.cpp:180-330, with its FormatError texts) as a table:
 * a line is `keyword words...`; a keyword's handler folds the line into a HeaderScan and returns an empty string or
 * the complaint.  Unknown keywords (comment, obj_info, ...) fold into nothing.
 */
using Words = std::vector<std::string>;

struct HeaderScan
{
    static constexpr const char *wanted[NUM_PROPERTIES] = {"x", "y", "z", "nx", "ny", "nz", "radius"};
    unsigned found = 0;                 /* bit i: wanted[i] has been declared */
    uint64_t declared = 0;              /* element lines so far; the properties that follow belong to the last one */
    bool format = false;
    uint64_t vertexCount = 0, vertexSize = 0;
    uint64_t offsets[NUM_PROPERTIES] = {};
};
constexpr const char *HeaderScan::wanted[NUM_PROPERTIES];

/* `property <type> <name>` or `property list <count type> <value type> <name>`, by word count */
struct PropertyDecl
{
    const TypeRow *value = nullptr, *length = nullptr;
    std::string name;

    std::string parse(const Words &w)
    {
        const bool list = w.size() >= 2 && w[1] == "list";
        if (w.size() != (list ? 5u : 3u))
            return "Malformed property line";
        const std::string *spell[2] = {&w[list ? 3 : 1], list ? &w[2] : nullptr};
        const TypeRow **slot[2] = {&value, &length};
        for (int k = 1; k >= 0; k--)                    /* the count type is complained about first */
            if (spell[k] != nullptr && (*slot[k] = findType(*spell[k])) == nullptr)
                return "Unknown type `" + *spell[k] + "'";
        if (length != nullptr && !length->integral)
            return "List cannot have floating-point count";
        name = w.back();
        return "";
    }
};

std::string foldFormat(HeaderScan &h, const Words &w)
{
    static const struct { const char *encoding, *complaint; } refused[] = {
        {"ascii", "PLY ASCII format not supported"},
        {"binary_big_endian", "PLY big endian format not supported on this CPU"},
    };
    if (w.size() != 3)
        return "Malformed format line";
    for (const auto &no : refused)
        if (w[1] == no.encoding)
            return no.complaint;
    if (w[1] != "binary_little_endian")
        return "Unknown PLY format " + w[1];
    if (w[2] != "1.0")
        return "Unknown PLY version " + w[2];
    h.format = true;
    return "";
}

std::string foldElement(HeaderScan &h, const Words &w)
{
    if (w.size() != 3)
        return "Malformed element line";
    /* a count is decimal digits that fit 64 bits, nothing else (boost::lexical_cast<size_type> in the reference) */
    uint64_t count = 0;
    bool fits = !w[2].empty() && w[2].find_first_not_of("0123456789") == std::string::npos;
    for (size_t i = 0; fits && i < w[2].size(); i++)
    {
        const uint64_t digit = (uint64_t) (w[2][i] - '0');
        fits = count <= (UINT64_MAX - digit) / 10;
        count = count * 10 + digit;
    }
    if (!fits)
        return "Malformed element line or too many elements";
    if (h.declared == 0)
    {
        if (w[1] != "vertex")
            return "First element is not vertex";
        h.vertexCount = count;
    }
    h.declared++;
    return "";
}

std::string foldProperty(HeaderScan &h, const Words &w)
{
    PropertyDecl p;
    const std::string bad = p.parse(w);
    if (!bad.empty())
        return bad;
    if (h.declared == 0)
        return "Property `" + p.name + "' appears before any element declaration";
    if (h.declared != 1)
        return "";                                      /* another element's: nothing to lay out */
    if (p.length != nullptr)
        return "Lists in a vertex are not supported";
    const int which = (int) (std::find(HeaderScan::wanted, HeaderScan::wanted + NUM_PROPERTIES, p.name) - HeaderScan::wanted);
    if (which < NUM_PROPERTIES)
    {
        if (h.found >> which & 1)
            return "Duplicate property " + p.name;
        if (p.value->type != FLOAT32)
            return "Property " + p.name + " must be FLOAT32";
        h.found |= 1u << which;
        h.offsets[which] = h.vertexSize;
    }
    h.vertexSize += p.value->bytes;
    return "";
}

const struct { const char *keyword; std::string (*fold)(HeaderScan &, const Words &); } headerLines[] = {
    {"format", foldFormat}, {"element", foldElement}, {"property", foldProperty},
};

int readHeader(mlsgpu_ply_reader &r)
{
    HeaderScan h;
    std::string line;
    if (!std::getline(r.in, line) || line != "ply")
        return formatError(r, "PLY signature missing");
    for (bool ended = false; !ended;)
    {
        if (!std::getline(r.in, line))
            return formatError(r, "End of file in PLY header");
        std::istringstream cut(line);
        const Words w{std::istream_iterator<std::string>(cut), std::istream_iterator<std::string>()};
        if (w.empty())
            continue;
        ended = w[0] == "end_header";
        for (const auto &row : headerLines)
            if (w[0] == row.keyword)
            {
                const std::string complaint = row.fold(h, w);
                if (!complaint.empty())
                    return formatError(r, complaint);
            }
    }
    if (!h.format)
        return formatError(r, "No format line found");
    if (h.declared == 0)
        return formatError(r, "No elements found");
    for (int i = 0; i < NUM_PROPERTIES; i++)
        if (!(h.found >> i & 1))
            return formatError(r, std::string("Property ") + HeaderScan::wanted[i] + " not found");
    r.vertexCount = h.vertexCount;
    r.vertexSize = h.vertexSize;
    std::copy(h.offsets, h.offsets + NUM_PROPERTIES, r.offsets);
    r.headerSize = (uint64_t) r.in.tellg();
    return MLSGPU_OK;
}

} // namespace

MLSGPU_API int mlsgpu_hip_ply_open(const char *path, float smooth, float maxRadius, mlsgpu_ply_reader **out)
{
    REQUIRE(path != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    std::unique_ptr<mlsgpu_ply_reader> r(new mlsgpu_ply_reader);
    r->path = path;
    r->smooth = smooth;
    r->maxRadius = maxRadius;
    r->in.open(path, std::ios::in | std::ios::binary);
    if (!r->in)
        return setError(MLSGPU_ERR_INVALID, "%s: could not open file", path);
    PROPAGATE(readHeader(*r));
    /* "File is too small to contain its vertices", src/fast_ply.cpp:352-372 */
    r->in.seekg(0, std::ios::end);
    const uint64_t fileSize = (uint64_t) r->in.tellg();
    if (r->vertexSize != 0 && (fileSize - r->headerSize) / r->vertexSize < r->vertexCount)
        return setError(MLSGPU_ERR_FORMAT, "%s: File is too small to contain its vertices", path);
    *out = r.release();
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_ply_close(mlsgpu_ply_reader *r) { delete r; }

MLSGPU_API uint64_t mlsgpu_hip_ply_size(const mlsgpu_ply_reader *r) { return r ? r->vertexCount : 0; }

MLSGPU_API int mlsgpu_hip_ply_layout(const mlsgpu_ply_reader *r, uint64_t out[10])
{
    REQUIRE(r != nullptr && out != nullptr, MLSGPU_ERR_INVALID);
    out[0] = r->vertexSize;
    out[1] = r->vertexCount;
    out[2] = r->headerSize;
    for (int i = 0; i < NUM_PROPERTIES; i++)
        out[3 + i] = r->offsets[i];
    return MLSGPU_OK;
}

/* Reader::Handle::read + Reader::decode, src/fast_ply.cpp:334-350, 374-400 */
MLSGPU_API int mlsgpu_hip_ply_read(mlsgpu_ply_reader *r, uint64_t first, uint64_t count, mlsgpu_splat *out)
{
    REQUIRE(r != nullptr && (count == 0 || out != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(first <= r->vertexCount && count <= r->vertexCount - first, MLSGPU_ERR_LENGTH);     /* std::out_of_range */
    const uint64_t batch = 1 << 16;
    r->buffer.resize((size_t) (std::min(batch, std::max<uint64_t>(count, 1)) * r->vertexSize));
    for (uint64_t done = 0; done < count; done += batch)
    {
        const uint64_t n = std::min(batch, count - done);
        r->in.clear();
        r->in.seekg((std::streamoff) (r->headerSize + (first + done) * r->vertexSize));
        r->in.read(r->buffer.data(), (std::streamsize) (n * r->vertexSize));
        if ((uint64_t) r->in.gcount() != n * r->vertexSize)
            return setError(MLSGPU_ERR_FORMAT, "%s: short read", r->path.c_str());
        for (uint64_t i = 0; i < n; i++)
        {
            const char *v = r->buffer.data() + i * r->vertexSize;
            mlsgpu_splat &s = out[done + i];
            std::memcpy(&s.position[0], v + r->offsets[X], 4);
            std::memcpy(&s.position[1], v + r->offsets[Y], 4);
            std::memcpy(&s.position[2], v + r->offsets[Z], 4);
            std::memcpy(&s.radius, v + r->offsets[RADIUS], 4);
            std::memcpy(&s.normal[0], v + r->offsets[NX], 4);
            std::memcpy(&s.normal[1], v + r->offsets[NY], 4);
            std::memcpy(&s.normal[2], v + r->offsets[NZ], 4);
            s.radius = std::min(s.radius, r->maxRadius);
            s.radius *= r->smooth;
            s.quality = (float) (1.0 / (s.radius * s.radius));       /* double 1.0 / float product, as the reference */
        }
    }
    return MLSGPU_OK;
}

namespace
{

/* decode `count` vertices starting at `first` into out, on `threads` host threads (each with its own file handle) */
int decodeParallel(mlsgpu_ply_reader *r, uint64_t first, uint64_t count, mlsgpu_splat *out, unsigned threads)
{
    if (threads <= 1 || count < 65536)
        return mlsgpu_hip_ply_read(r, first, count, out);
    std::vector<std::thread> pool;
    std::vector<int> rc(threads, MLSGPU_OK);
    std::vector<std::string> msg(threads);
    const uint64_t per = (count + threads - 1) / threads;
    for (unsigned t = 0; t < threads; t++)
    {
        const uint64_t lo = std::min<uint64_t>(count, t * per), hi = std::min<uint64_t>(count, lo + per);
        if (lo == hi)
            continue;
        pool.emplace_back([=, &rc, &msg]
        {
            mlsgpu_ply_reader *mine = nullptr;
            rc[t] = mlsgpu_hip_ply_open(r->path.c_str(), r->smooth, r->maxRadius, &mine);
            if (rc[t] == MLSGPU_OK)
                rc[t] = mlsgpu_hip_ply_read(mine, first + lo, hi - lo, out + lo);
            if (rc[t] != MLSGPU_OK)
                msg[t] = mlsgpu_hip_last_error();
            mlsgpu_hip_ply_close(mine);
        });
    }
    for (std::thread &t : pool)
        t.join();
    for (unsigned t = 0; t < threads; t++)
        if (rc[t] != MLSGPU_OK)
            return setError(rc[t], "%s", msg[t].c_str());
    return MLSGPU_OK;
}

} // namespace

/* File -> device without a host copy of the whole cloud: the role of the reference's reader threads and async I/O
 * (src/splat_set.h:389-700, src/async_io.h) for inputs that fit in HBM.  Two pinned buffers: the host threads decode
 * batch k + 1 from the file while batch k travels over PCIe on the context's stream. */
MLSGPU_API int mlsgpu_hip_ply_load(mlsgpu_ply_reader *r, mlsgpu_ctx *ctx, uint64_t first, uint64_t count, mlsgpu_splat *dOut,
                                   uint32_t hostThreads)
{
    REQUIRE(r != nullptr && ctx != nullptr && (count == 0 || dOut != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(first <= r->vertexCount && count <= r->vertexCount - first, MLSGPU_ERR_LENGTH);
    if (count == 0)
        return MLSGPU_OK;
    HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t batch = std::min<uint64_t>(count, uint64_t(1) << 21);        /* 64 MiB of splats */
    mlsgpu_splat *pinned[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = MLSGPU_OK;
    for (int b = 0; b < 2 && rc == MLSGPU_OK; b++)
        if (hipHostMalloc((void **) &pinned[b], batch * sizeof(mlsgpu_splat)) != hipSuccess
            || hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess)
            rc = setError(MLSGPU_ERR_NOMEM, "ply load: cannot allocate the pinned buffers");
    bool busy[2] = {false, false};
    int cur = 0;
    for (uint64_t at = 0; at < count && rc == MLSGPU_OK; at += batch)
    {
        const uint64_t n = std::min(batch, count - at);
        if (busy[cur] && hipEventSynchronize(done[cur]) != hipSuccess)
            rc = setError(MLSGPU_ERR_HIP, "ply load: waiting for the previous copy failed");
        if (rc == MLSGPU_OK)
            rc = decodeParallel(r, first + at, n, pinned[cur], hostThreads == 0 ? 4u : hostThreads);
        if (rc == MLSGPU_OK
            && (hipMemcpyAsync(dOut + at, pinned[cur], n * sizeof(mlsgpu_splat), hipMemcpyHostToDevice, ctx->stream) != hipSuccess
                || hipEventRecord(done[cur], ctx->stream) != hipSuccess))
            rc = setError(MLSGPU_ERR_HIP, "ply load: the host-to-device copy failed");
        busy[cur] = true;
        cur ^= 1;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MLSGPU_OK)
        rc = setError(MLSGPU_ERR_HIP, "ply load: synchronise failed");
    for (int b = 0; b < 2; b++)
    {
        if (pinned[b]) hipHostFree(pinned[b]);
        if (done[b]) hipEventDestroy(done[b]);
    }
    return rc;
}

/* ------------------------------------------------------------------ SplatSet::FileSet, src/splat_set.h:383-700 */

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <fcntl.h>
#include <unistd.h>

/* where a file keeps what a splat needs: enough to decode its rows without the reader */
struct FileLayout
{
    uint64_t vertexSize = 0, headerSize = 0;
    uint64_t offsets[NUM_PROPERTIES] = {0, 0, 0, 0, 0, 0, 0};
    /* rows of whole 32-bit words, no wider than a splat: they can cross the link as they are and be decoded on the device */
    bool wordRows() const
    {
        bool ok = vertexSize % 4 == 0 && vertexSize <= sizeof(mlsgpu_splat);
        for (int i = 0; i < NUM_PROPERTIES; i++)
            ok = ok && offsets[i] % 4 == 0;
        return ok;
    }
};

struct mlsgpu_fileset
{
    float smooth = 1.0f, maxRadius = 0.0f;
    std::vector<std::string> paths;
    std::vector<uint64_t> first;            /* first[i] = linear id of file i's first splat; first[n] = total */
    std::vector<FileLayout> layouts;
    uint64_t bufferSize = 32u << 20;        /* FileSet::DEFAULT_BUFFER_SIZE, src/splat_set.h:461 */
    /* mlsgpu_hip_fileset_load's staging, kept from call to call (pinning half a gigabyte costs as much as loading 10^8
     * splats; mlsgpu_hip_bucket_stream loads chunk after chunk) */
    std::mutex loadMutex;
    char *pinned = nullptr, *dRaw = nullptr;
    uint64_t pinnedBytes = 0, rawBytes = 0;
    int stagingDevice = -1;
    void dropStaging()
    {
        if (pinned) (void) hipHostFree(pinned);
        if (dRaw) (void) hipFree(dRaw);
        pinned = dRaw = nullptr;
        pinnedBytes = rawBytes = 0;
    }
    ~mlsgpu_fileset() { dropStaging(); }
};

MLSGPU_API int mlsgpu_hip_fileset_create(float smooth, float maxRadius, mlsgpu_fileset **out)
{
    REQUIRE(out != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_fileset *f = new mlsgpu_fileset;
    f->smooth = smooth;
    f->maxRadius = maxRadius;
    f->first.push_back(0);
    *out = f;
    return MLSGPU_OK;
}

MLSGPU_API void mlsgpu_hip_fileset_destroy(mlsgpu_fileset *f) { delete f; }

/* FileSet::addFile, src/splat_set_impl.h: the header is checked now, the file is re-opened by whoever reads it */
MLSGPU_API int mlsgpu_hip_fileset_add_file(mlsgpu_fileset *f, const char *path)
{
    REQUIRE(f != nullptr && path != nullptr, MLSGPU_ERR_INVALID);
    mlsgpu_ply_reader *r = nullptr;
    PROPAGATE(mlsgpu_hip_ply_open(path, f->smooth, f->maxRadius, &r));
    const uint64_t n = r->vertexCount;
    FileLayout layout;
    layout.vertexSize = r->vertexSize;
    layout.headerSize = r->headerSize;
    for (int i = 0; i < NUM_PROPERTIES; i++)
        layout.offsets[i] = r->offsets[i];
    mlsgpu_hip_ply_close(r);
    f->layouts.push_back(layout);
    f->paths.push_back(path);
    f->first.push_back(f->first.back() + n);
    return MLSGPU_OK;
}

MLSGPU_API uint64_t mlsgpu_hip_fileset_num_files(const mlsgpu_fileset *f) { return f ? f->paths.size() : 0; }
MLSGPU_API uint64_t mlsgpu_hip_fileset_num_splats(const mlsgpu_fileset *f) { return f ? f->first.back() : 0; }

/* FileSet::setBufferSize, src/splat_set.h:505-515: the memory a stream may pin; a quarter of it per read */
MLSGPU_API int mlsgpu_hip_fileset_set_buffer_size(mlsgpu_fileset *f, uint64_t bytes)
{
    REQUIRE(f != nullptr && bytes >= 4 * sizeof(mlsgpu_splat), MLSGPU_ERR_INVALID);
    f->bufferSize = bytes;
    return MLSGPU_OK;
}

namespace
{

/* splats [first, first + count) of the concatenation, file by file (FileRangeIterator: ranges "always within a single
 * file", src/splat_set.h:399-420); `readers` caches one open reader per file for the calling thread */
int filesetRead(const mlsgpu_fileset *f, std::vector<mlsgpu_ply_reader *> &readers, uint64_t first, uint64_t count,
                mlsgpu_splat *out)
{
    size_t file = std::upper_bound(f->first.begin(), f->first.end(), first) - f->first.begin() - 1;
    while (count > 0)
    {
        while (file + 1 < f->first.size() && f->first[file + 1] <= first)
            file++;
        const uint64_t inFile = first - f->first[file];
        const uint64_t n = std::min(count, f->first[file + 1] - first);
        if (readers[file] == nullptr)
        {
            for (mlsgpu_ply_reader *&other : readers)       /* one open file per caller at a time */
                if (other != nullptr)
                {
                    mlsgpu_hip_ply_close(other);
                    other = nullptr;
                }
            PROPAGATE(mlsgpu_hip_ply_open(f->paths[file].c_str(), f->smooth, f->maxRadius, &readers[file]));
        }
        PROPAGATE(mlsgpu_hip_ply_read(readers[file], inFile, n, out));
        out += n;
        first += n;
        count -= n;
    }
    return MLSGPU_OK;
}

void closeAll(std::vector<mlsgpu_ply_reader *> &readers)
{
    for (mlsgpu_ply_reader *r : readers)
        mlsgpu_hip_ply_close(r);
}

} // namespace

MLSGPU_API int mlsgpu_hip_fileset_read(mlsgpu_fileset *f, uint64_t first, uint64_t count, mlsgpu_splat *out)
{
    REQUIRE(f != nullptr && (count == 0 || out != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(first <= f->first.back() && count <= f->first.back() - first, MLSGPU_ERR_LENGTH);
    std::vector<mlsgpu_ply_reader *> readers(f->paths.size(), nullptr);
    const int rc = filesetRead(f, readers, first, count, out);
    closeAll(readers);
    return rc;
}

/*
 * Files -> device memory with bounded host memory: what the reference's ReaderThread + circular buffer + async I/O do
 * for its out-of-core splat sets (src/splat_set.h:560-700, src/async_io.h:95-140), for clouds that fit in HBM (one
 * billion splats are 32 GB of 288).  `readerThreads` host threads (0 = 4, at most 64) decode consecutive chunks of
 * bufferSize / max(4, readerThreads) bytes, each into its own slot of ONE pinned buffer of bufferSize bytes; the calling thread sends every finished
 * chunk to dOut on ctx's stream, and a quarter is reused once its copy has completed -- file reads, decoding and PCIe
 * overlap, and the host never holds more than bufferSize bytes of the cloud however many files of whatever size.
 */
namespace
{

/* Reader::decode (src/fast_ply.cpp:374-400) on the device, for rows that crossed the link as the file holds them */
struct RowLayout
{
    uint32_t words;                     /* 32-bit words per row */
    uint32_t at[NUM_PROPERTIES];        /* word of each property inside a row */
    float smooth, maxRadius;
};

__global__ __launch_bounds__(256) void decodeRowsKernel(const uint32_t *rows, uint64_t n, RowLayout R, mlsgpu_splat *out)
{
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t *const v = rows + i * R.words;
    float r = __uint_as_float(v[R.at[RADIUS]]);
    r = R.maxRadius < r ? R.maxRadius : r;                /* std::min(radius, maxRadius): a NaN radius stays */
    r *= R.smooth;
    const float q = (float) (1.0 / (double) (r * r));       /* double 1.0 / float product, as the reference */
    float4 *const o = reinterpret_cast<float4 *>(out + i);
    o[0] = make_float4(__uint_as_float(v[R.at[X]]), __uint_as_float(v[R.at[Y]]), __uint_as_float(v[R.at[Z]]), r);
    o[1] = make_float4(__uint_as_float(v[R.at[NX]]), __uint_as_float(v[R.at[NY]]), __uint_as_float(v[R.at[NZ]]), q);
}

/* a run of rows of ONE file inside a job */
struct RowPiece
{
    size_t file;
    uint64_t firstRow, rows, byteAt;    /* byteAt: where the run starts in the job's slot */
};

void piecesOf(const mlsgpu_fileset *f, uint64_t first, uint64_t count, std::vector<RowPiece> &out)
{
    out.clear();
    size_t file = std::upper_bound(f->first.begin(), f->first.end(), first) - f->first.begin() - 1;
    uint64_t byteAt = 0;
    while (count > 0)
    {
        while (file + 1 < f->first.size() && f->first[file + 1] <= first)
            file++;
        const uint64_t n = std::min(count, f->first[file + 1] - first);
        out.push_back(RowPiece{file, first - f->first[file], n, byteAt});
        byteAt += n * f->layouts[file].vertexSize;
        first += n;
        count -= n;
    }
}

bool rawRowsWanted()
{
    const char *e = getenv("MLSGPU_HIP_FILESET_RAW");
    return e == nullptr || e[0] != '0';
}

} // namespace

/*
 * Two routes per job (chunk of consecutive splats):
 *   rows     (every file's rows are whole words, no wider than a splat -- x y z nx ny nz radius as float32 is 28 bytes)
 *            a reader thread only READS the rows into its pinned slot (pread: one copy by the kernel, no pass of its own
 *            over the bytes); the rows cross the link as they are and decodeRowsKernel writes the splats: the host touches
 *            28 bytes per splat once instead of reading 28 and writing 32 behind a scalar decode, the link carries 28
 *            instead of 32;
 *   splats   (any other layout, or MLSGPU_HIP_FILESET_RAW=0) the reader threads decode on the host, as Reader::decode.
 * Same splats either way, bit for bit (tests/test_fileset.py).
 */
MLSGPU_API int mlsgpu_hip_fileset_load(mlsgpu_fileset *f, mlsgpu_ctx *ctx, uint64_t first, uint64_t count, mlsgpu_splat *dOut,
                                       uint32_t readerThreads)
{
    REQUIRE(f != nullptr && ctx != nullptr && (count == 0 || dOut != nullptr), MLSGPU_ERR_INVALID);
    REQUIRE(first <= f->first.back() && count <= f->first.back() - first, MLSGPU_ERR_LENGTH);
    if (count == 0)
        return MLSGPU_OK;
    std::lock_guard<std::mutex> oneLoad(f->loadMutex);
    HIP_CHECK(hipSetDevice(ctx->device));
    /* one slot per reader thread, at least four (the reference pipelines reads through a fixed fraction of its buffer,
     * src/splat_set.h:455-462); a host with many cores decodes with as many threads as the caller asks for */
    const uint32_t wanted = std::min<uint32_t>(readerThreads == 0 ? 4u : readerThreads, 64u);
    const uint64_t SLOTS = std::max<uint32_t>(4u, wanted);
    const uint64_t chunk = std::max<uint64_t>(1, f->bufferSize / SLOTS / sizeof(mlsgpu_splat));
    const uint64_t jobs = (count + chunk - 1) / chunk;
    const uint32_t threads = (uint32_t) std::min<uint64_t>(wanted, jobs);
    bool rawRows = rawRowsWanted();
    for (const FileLayout &l : f->layouts)
        rawRows = rawRows && l.wordRows();
    const uint64_t slotBytes = chunk * sizeof(mlsgpu_splat);
    if (f->stagingDevice != ctx->device || f->pinnedBytes < SLOTS * slotBytes)
    {
        f->dropStaging();
        f->stagingDevice = ctx->device;
        if (hipHostMalloc((void **) &f->pinned, SLOTS * slotBytes) != hipSuccess)
        {
            f->pinned = nullptr;
            return setError(MLSGPU_ERR_NOMEM, "fileset load: cannot pin %llu bytes", (unsigned long long) (SLOTS * slotBytes));
        }
        f->pinnedBytes = SLOTS * slotBytes;
    }
    if (rawRows && f->rawBytes < SLOTS * slotBytes)
    {
        if (f->dRaw) (void) hipFree(f->dRaw);
        f->dRaw = nullptr;
        f->rawBytes = 0;
        if (hipMalloc((void **) &f->dRaw, SLOTS * slotBytes) == hipSuccess)
            f->rawBytes = SLOTS * slotBytes;
        else
        {
            (void) hipGetLastError();
            rawRows = false;                /* no room for the rows on the device: decode on the host */
        }
    }
    char *const pinned = f->pinned;
    std::vector<hipEvent_t> copied(SLOTS, nullptr);
    int rc = MLSGPU_OK;
    for (uint64_t s = 0; s < SLOTS && rc == MLSGPU_OK; s++)
        if (hipEventCreateWithFlags(&copied[s], hipEventDisableTiming) != hipSuccess)
            rc = setError(MLSGPU_ERR_HIP, "fileset load: cannot create an event");

    std::mutex mutex;
    std::condition_variable cond;
    std::vector<char> ready(jobs, 0);           /* job decoded (or read) into its slot */
    uint64_t issued = 0;                         /* jobs whose copy has been enqueued (by the calling thread, in order) */
    std::atomic<uint64_t> next(0);
    int failed = MLSGPU_OK;
    std::string failText;

    auto readRows = [&](std::vector<int> &fds, uint64_t at, uint64_t n, char *slot) -> int
    {
        std::vector<RowPiece> pieces;
        piecesOf(f, first + at, n, pieces);
        for (const RowPiece &p : pieces)
        {
            const FileLayout &l = f->layouts[p.file];
            if (fds[p.file] < 0)
            {
                /* a thread walks the files in order: one descriptor each at a time, however many files the set has */
                for (int &fd : fds)
                    if (fd >= 0)
                    {
                        close(fd);
                        fd = -1;
                    }
                fds[p.file] = open(f->paths[p.file].c_str(), O_RDONLY | O_CLOEXEC);
                if (fds[p.file] < 0)
                    return setError(MLSGPU_ERR_INVALID, "%s: could not open file", f->paths[p.file].c_str());
            }
            uint64_t done = 0;
            const uint64_t bytes = p.rows * l.vertexSize;
            while (done < bytes)
            {
                const ssize_t got = pread(fds[p.file], slot + p.byteAt + done, bytes - done,
                                          (off_t) (l.headerSize + p.firstRow * l.vertexSize + done));
                if (got <= 0)
                    return setError(MLSGPU_ERR_FORMAT, "%s: short read", f->paths[p.file].c_str());
                done += (uint64_t) got;
            }
        }
        return MLSGPU_OK;
    };

    /* the readers run -- and first touch the pinned slots -- on the GPU's NUMA node, whatever the caller is bound to:
     * the link reads the slots at its own rate only from there (csrc/placement.hpp) */
    const std::vector<int> nodeCpus = placement::cpusOfNode(mlsgpu_hip_device_node(ctx->device));
    auto readerMain = [&]()
    {
        placement::bindThisThread(nodeCpus);
        std::vector<mlsgpu_ply_reader *> readers(f->paths.size(), nullptr);
        std::vector<int> fds(f->paths.size(), -1);
        for (;;)
        {
            const uint64_t j = next.fetch_add(1);
            if (j >= jobs)
                break;
            {
                /* slot j % SLOTS is free once the copy of job j - SLOTS has been enqueued and has completed */
                std::unique_lock<std::mutex> l(mutex);
                cond.wait(l, [&] { return failed != MLSGPU_OK || j < SLOTS || issued > j - SLOTS; });
                if (failed != MLSGPU_OK)
                    break;
            }
            int r = MLSGPU_OK;
            if (j >= SLOTS && hipEventSynchronize(copied[j % SLOTS]) != hipSuccess)
                r = setError(MLSGPU_ERR_HIP, "fileset load: waiting for a copy failed");
            const uint64_t at = j * chunk, n = std::min(chunk, count - at);
            char *const slot = pinned + (j % SLOTS) * slotBytes;
            if (r == MLSGPU_OK)
                r = rawRows ? readRows(fds, at, n, slot) : filesetRead(f, readers, first + at, n, reinterpret_cast<mlsgpu_splat *>(slot));
            std::lock_guard<std::mutex> l(mutex);
            if (r != MLSGPU_OK && failed == MLSGPU_OK)
            {
                failed = r;
                failText = mlsgpu_hip_last_error();
            }
            ready[j] = 1;
            cond.notify_all();
            if (r != MLSGPU_OK)
                break;
        }
        closeAll(readers);
        for (int fd : fds)
            if (fd >= 0)
                close(fd);
    };
    std::vector<std::thread> pool;
    if (rc == MLSGPU_OK)
        for (uint32_t t = 0; t < threads; t++)
            pool.emplace_back(readerMain);
    std::vector<RowPiece> pieces;
    for (uint64_t j = 0; j < jobs && rc == MLSGPU_OK; j++)
    {
        {
            std::unique_lock<std::mutex> l(mutex);
            cond.wait(l, [&] { return failed != MLSGPU_OK || ready[j]; });
            if (failed != MLSGPU_OK)
                break;
        }
        const uint64_t at = j * chunk, n = std::min(chunk, count - at);
        const char *const slot = pinned + (j % SLOTS) * slotBytes;
        hipError_t e = hipSuccess;
        if (rawRows)
        {
            piecesOf(f, first + at, n, pieces);
            const uint64_t bytes = pieces.back().byteAt + pieces.back().rows * f->layouts[pieces.back().file].vertexSize;
            char *const dSlot = f->dRaw + (j % SLOTS) * slotBytes;
            e = hipMemcpyAsync(dSlot, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
            uint64_t outAt = at;
            for (const RowPiece &p : pieces)
            {
                if (e != hipSuccess)
                    break;
                const FileLayout &l = f->layouts[p.file];
                RowLayout R;
                R.words = (uint32_t) (l.vertexSize / 4);
                for (int i = 0; i < NUM_PROPERTIES; i++)
                    R.at[i] = (uint32_t) (l.offsets[i] / 4);
                R.smooth = f->smooth;
                R.maxRadius = f->maxRadius;
                hipLaunchKernelGGL(decodeRowsKernel, dim3((uint32_t) ((p.rows + 255) / 256)), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<const uint32_t *>(dSlot + p.byteAt), p.rows, R, dOut + outAt);
                e = hipGetLastError();
                outAt += p.rows;
            }
        }
        else
            e = hipMemcpyAsync(dOut + at, slot, n * sizeof(mlsgpu_splat), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipEventRecord(copied[j % SLOTS], ctx->stream);
        std::lock_guard<std::mutex> l(mutex);
        if (e != hipSuccess && failed == MLSGPU_OK)
        {
            failed = MLSGPU_ERR_HIP;
            failText = std::string("fileset load: the host-to-device copy failed: ") + hipGetErrorString(e);
        }
        issued = j + 1;
        cond.notify_all();
    }
    {
        std::lock_guard<std::mutex> l(mutex);
        if (rc != MLSGPU_OK && failed == MLSGPU_OK)
            failed = rc;
        cond.notify_all();
    }
    for (std::thread &t : pool)
        t.join();
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && failed == MLSGPU_OK)
    {
        failed = MLSGPU_ERR_HIP;
        failText = "fileset load: synchronise failed";
    }
    for (uint64_t s = 0; s < SLOTS; s++)
        if (copied[s]) hipEventDestroy(copied[s]);
    if (failed != MLSGPU_OK)
        return setError(failed, "%s", failText.empty() ? mlsgpu_hip_last_error() : failText.c_str());
    return MLSGPU_OK;
}
