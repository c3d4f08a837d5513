// ntk_fastx_codecs.cpp - bzip2 / xz / zstd input for the record reader.  OUTSIDE SURVEY.md section 8 (its row f3 is gzip only: "only gzip
// matters"); kept because the reference reads these containers transparently when built with its "compression" feature
// (src/parser/mod.rs:27-35,109-147) and the reader's conformance tests hold such files.  The image ships the codecs' run-time libraries but
// not their headers, so the few entry points and structs of their stable C ABIs are declared here and the libraries are loaded on first
// use.  The library builds without this file under -DNTK_NO_EXTRA_CODECS (the reader then reports such input as unreadable).
#include "ntk_fastx.hpp"

#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

namespace ntk {

#ifndef NTK_NO_EXTRA_CODECS
namespace {

struct Bz2Stream {   // bz_stream, bzlib.h 1.0
    char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
struct Bz2Decoder : StreamDecoder {
    Bz2Stream s; int (*dec)(Bz2Stream *) = nullptr; int (*end_)(Bz2Stream *) = nullptr; bool ok = false;
    Bz2Decoder()
    {
        memset(&s, 0, sizeof(s));
        void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        auto init = (int (*)(Bz2Stream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
        dec = (int (*)(Bz2Stream *))dlsym(h, "BZ2_bzDecompress");
        end_ = (int (*)(Bz2Stream *))dlsym(h, "BZ2_bzDecompressEnd");
        ok = init && dec && end_ && init(&s, 0, 0) == 0;
    }
    ~Bz2Decoder() override { if (ok) end_(&s); }
    bool step(const uint8_t *in, size_t in_n, size_t *used, uint8_t *out, size_t cap, size_t *made, bool *end) override
    {
        s.next_in = (char *)in; s.avail_in = (unsigned)(in_n > 0x40000000u ? 0x40000000u : in_n);
        s.next_out = (char *)out; s.avail_out = (unsigned)(cap > 0x40000000u ? 0x40000000u : cap);
        const unsigned in0 = s.avail_in, out0 = s.avail_out;
        const int rc = dec(&s);
        *used = in0 - s.avail_in; *made = out0 - s.avail_out;
        *end = rc == 4;  // BZ_STREAM_END
        return rc == 0 || rc == 4;
    }
    const char *name() const override { return "bzip2"; }
};

struct LzmaStream {  // lzma_stream, lzma/base.h 5.x (LZMA_STREAM_INIT is all zeros)
    const uint8_t *next_in; size_t avail_in; uint64_t total_in;
    uint8_t *next_out; size_t avail_out; uint64_t total_out;
    const void *allocator; void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
struct XzDecoder : StreamDecoder {
    LzmaStream s; int (*code)(LzmaStream *, int) = nullptr; void (*end_)(LzmaStream *) = nullptr; bool ok = false;
    XzDecoder()
    {
        memset(&s, 0, sizeof(s));
        void *h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        auto init = (int (*)(LzmaStream *, uint64_t, uint32_t))dlsym(h, "lzma_stream_decoder");
        code = (int (*)(LzmaStream *, int))dlsym(h, "lzma_code");
        end_ = (void (*)(LzmaStream *))dlsym(h, "lzma_end");
        ok = init && code && end_ && init(&s, UINT64_MAX, 0) == 0;
    }
    ~XzDecoder() override { if (ok) end_(&s); }
    bool step(const uint8_t *in, size_t in_n, size_t *used, uint8_t *out, size_t cap, size_t *made, bool *end) override
    {
        s.next_in = in; s.avail_in = in_n; s.next_out = out; s.avail_out = cap;
        const int rc = code(&s, 0 /* LZMA_RUN */);
        *used = in_n - s.avail_in; *made = cap - s.avail_out;
        *end = rc == 1;  // LZMA_STREAM_END
        return rc == 0 || rc == 1 || (rc == 10 /* LZMA_BUF_ERROR: no progress possible yet */ && *used == 0 && *made == 0);
    }
    const char *name() const override { return "xz"; }
};

struct ZstdBuf { void *p; size_t size, pos; };  // ZSTD_inBuffer / ZSTD_outBuffer, zstd.h 1.x
struct ZstdDecoder : StreamDecoder {
    void *ds = nullptr; size_t (*dec)(void *, ZstdBuf *, ZstdBuf *) = nullptr; size_t (*free_)(void *) = nullptr;
    unsigned (*is_err)(size_t) = nullptr; bool ok = false;
    ZstdDecoder()
    {
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        auto create = (void *(*)())dlsym(h, "ZSTD_createDStream");
        dec = (size_t (*)(void *, ZstdBuf *, ZstdBuf *))dlsym(h, "ZSTD_decompressStream");
        free_ = (size_t (*)(void *))dlsym(h, "ZSTD_freeDStream");
        is_err = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
        if (create && dec && free_ && is_err) ds = create();
        ok = ds != nullptr;
    }
    ~ZstdDecoder() override { if (ds) free_(ds); }
    bool step(const uint8_t *in, size_t in_n, size_t *used, uint8_t *out, size_t cap, size_t *made, bool *end) override
    {
        ZstdBuf ib{(void *)in, in_n, 0}, ob{out, cap, 0};
        const size_t rc = dec(ds, &ob, &ib);
        *used = ib.pos; *made = ob.pos;
        *end = false;   // frames may follow one another (the zstd crate's Decoder reads them all): the input's end ends it
        // 0: a frame has been decoded and flushed completely (a call that neither consumed nor produced anything - the
        // probe at the input's end - says nothing about the frame before it)
        if (!is_err(rc) && (ib.pos > 0 || ob.pos > 0)) in_frame = rc != 0;
        return !is_err(rc);
    }
    bool mid_stream() const override { return in_frame; }
    bool in_frame = false;
    const char *name() const override { return "zstd"; }
};

}  // namespace

StreamDecoder *make_stream_decoder(uint8_t first_magic_byte)
{
    if (first_magic_byte == 0x42) { auto *d = new Bz2Decoder(); if (d->ok) return d; delete d; return nullptr; }
    if (first_magic_byte == 0xFD) { auto *d = new XzDecoder(); if (d->ok) return d; delete d; return nullptr; }
    auto *d = new ZstdDecoder();
    if (d->ok) return d;
    delete d;
    return nullptr;
}
#else
StreamDecoder *make_stream_decoder(uint8_t) { return nullptr; }
#endif

}  // namespace ntk
