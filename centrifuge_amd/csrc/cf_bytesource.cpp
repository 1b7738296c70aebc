// cf_bytesource.cpp — see cf_bytesource.hpp.
#include "cf_bytesource.hpp"
#include <sys/stat.h>

#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

extern char **environ;

namespace cfamd {

struct ByteSource::Impl {
    virtual ~Impl() = default;
    virtual size_t read(char *dst, size_t n) = 0;
    virtual int fd() const { return -1; }
};

namespace {

bool endsWith(const std::string &s, const char *suf) {
    const size_t n = std::char_traits<char>::length(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

[[noreturn]] void cannotOpen(const std::string &path) {
    throw std::runtime_error("Warning: Could not open read file \"" + path + "\" for reading");
}

struct PlainImpl : ByteSource::Impl {
    std::FILE *f;
    bool own;
    std::string path;
    PlainImpl(std::FILE *f_, bool own_, std::string p) : f(f_), own(own_), path(std::move(p)) {}
    ~PlainImpl() override { if (own && f) std::fclose(f); }
    int fd() const override { return own ? fileno(f) : -1; }
    size_t read(char *dst, size_t n) override {
        const size_t got = std::fread(dst, 1, n, f);
        if (got < n && std::ferror(f)) throw std::runtime_error("Error: I/O error while reading \"" + path + "\"");
        return got;
    }
};

// A gzip file as a stream of deflate members (RFC 1952), inflated with zlib as it is read.
struct GzImpl : ByteSource::Impl {
    std::FILE *f;
    std::string path;
    z_stream z{};
    std::vector<unsigned char> in;
    bool zInit = false, fileEof = false, memberOpen = false, sawMember = false;
    GzImpl(std::FILE *f_, std::string p) : f(f_), path(std::move(p)), in(1u << 20) {
        if (inflateInit2(&z, 15 + 16) != Z_OK) { std::fclose(f); throw std::runtime_error("zlib: inflateInit2 failed"); }
        zInit = true;
    }
    ~GzImpl() override { if (zInit) inflateEnd(&z); if (f) std::fclose(f); }
    size_t read(char *dst, size_t n) override {
        size_t done = 0;
        while (done < n) {
            if (z.avail_in == 0 && !fileEof) {
                const size_t got = std::fread(in.data(), 1, in.size(), f);
                if (got < in.size()) { if (std::ferror(f)) throw std::runtime_error("Error: I/O error while reading \"" + path + "\""); fileEof = true; }
                z.next_in = in.data(); z.avail_in = (uInt)got;
            }
            if (z.avail_in == 0 && fileEof) {
                if (memberOpen || !sawMember) throw std::runtime_error("Error: \"" + path + "\" is truncated or not a gzip file");
                break;                                                   // clean end after a complete member
            }
            if (!memberOpen) {
                // trailing zero padding after the last member is tolerated (as gzip does)
                while (z.avail_in && sawMember && *z.next_in == 0) { z.next_in++; z.avail_in--; }
                if (z.avail_in == 0) continue;
                memberOpen = true;
            }
            const size_t want = std::min<size_t>(n - done, 1u << 30);
            z.next_out = reinterpret_cast<Bytef *>(dst + done); z.avail_out = (uInt)want;
            const int rc = inflate(&z, Z_NO_FLUSH);
            done += want - z.avail_out;
            if (rc == Z_STREAM_END) {
                memberOpen = false; sawMember = true;
                if (inflateReset(&z) != Z_OK) throw std::runtime_error("zlib: inflateReset failed");
            } else if (rc != Z_OK && rc != Z_BUF_ERROR)
                throw std::runtime_error("Error: \"" + path + "\" is not a valid gzip file (" + (z.msg ? z.msg : "inflate error") + ")");
            else if (rc == Z_BUF_ERROR && z.avail_in == 0 && fileEof)
                throw std::runtime_error("Error: \"" + path + "\" is truncated");
        }
        return done;
    }
};

// BGZF (bgzip / htslib): a gzip file whose members are independent blocks of <= 64 KiB that carry their own
// compressed size in a 'BC' extra field — a batch of blocks is inflated on several threads.
struct BgzfImpl : ByteSource::Impl {
    static constexpr size_t kBatch = 512, kSlot = 65536;
    std::FILE *f;
    std::string path;
    int threads;
    std::vector<unsigned char> comp;                  // the batch's blocks back to back
    std::vector<size_t> cOff, cLen, uLen;
    std::vector<char> out;                            // kBatch slots of 64 KiB
    size_t curBlock = 0, curPos = 0, nBlocks = 0;
    bool eof = false;
    BgzfImpl(std::FILE *f_, std::string p, int t) : f(f_), path(std::move(p)), threads(std::max(1, t)), out(kBatch * kSlot) {}
    ~BgzfImpl() override { if (f) std::fclose(f); }

    [[noreturn]] void bad(const char *what) { throw std::runtime_error("Error: \"" + path + "\": " + what); }

    static void inflateBlock(const unsigned char *src, size_t n, char *dst, size_t expect, bool &ok) {
        // src = one whole gzip member; raw deflate data sits between the header (12 + XLEN bytes) and the 8-byte trailer
        const size_t xlen = src[10] | (src[11] << 8), hdr = 12 + xlen;
        z_stream z{};
        ok = false;
        if (n < hdr + 8 || inflateInit2(&z, -15) != Z_OK) return;
        z.next_in = const_cast<Bytef *>(src + hdr); z.avail_in = (uInt)(n - hdr - 8);
        z.next_out = reinterpret_cast<Bytef *>(dst); z.avail_out = (uInt)expect;
        const int rc = inflate(&z, Z_FINISH);
        ok = rc == Z_STREAM_END && z.avail_out == 0;
        if (ok && expect) {
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(dst), (uInt)expect);
            uint32_t want; std::memcpy(&want, src + n - 8, 4);
            ok = crc == want;
        }
        inflateEnd(&z);
    }

    void fill() {
        comp.clear(); cOff.clear(); cLen.clear(); uLen.clear();
        nBlocks = curBlock = curPos = 0;
        while (nBlocks < kBatch && !eof) {
            unsigned char h[18];
            const size_t got = std::fread(h, 1, 18, f);
            if (got == 0 && !std::ferror(f)) { eof = true; break; }
            if (got < 18) bad("truncated BGZF block header");
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) bad("not a BGZF block");
            const size_t xlen = h[10] | (h[11] << 8);
            if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) bad("BGZF block without a leading BC field");
            const size_t bsize = (size_t)(h[16] | (h[17] << 8)) + 1;
            if (bsize < 12 + xlen + 8) bad("bad BGZF block size");
            const size_t o = comp.size();
            comp.resize(o + bsize);
            std::memcpy(comp.data() + o, h, 18);
            if (std::fread(comp.data() + o + 18, 1, bsize - 18, f) != bsize - 18) bad("truncated BGZF block");
            uint32_t isz; std::memcpy(&isz, comp.data() + o + bsize - 4, 4);
            if (isz > kSlot) bad("BGZF block larger than 64 KiB");
            cOff.push_back(o); cLen.push_back(bsize); uLen.push_back(isz);
            nBlocks++;
        }
        if (nBlocks == 0) return;
        const int nt = (int)std::min<size_t>((size_t)threads, nBlocks);
        std::vector<char> okv(nBlocks, 0);
        auto work = [&](int t) {
            for (size_t b = (size_t)t; b < nBlocks; b += (size_t)nt) {
                bool ok; inflateBlock(comp.data() + cOff[b], cLen[b], out.data() + b * kSlot, uLen[b], ok);
                okv[b] = ok ? 1 : 0;
            }
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (size_t b = 0; b < nBlocks; b++) if (!okv[b]) bad("corrupt BGZF block");
    }

    size_t read(char *dst, size_t n) override {
        size_t done = 0;
        while (done < n) {
            if (curBlock >= nBlocks) {
                if (eof) break;
                fill();
                if (nBlocks == 0) break;
            }
            const size_t avail = uLen[curBlock] - curPos;
            const size_t take = std::min(avail, n - done);
            std::memcpy(dst + done, out.data() + curBlock * kSlot + curPos, take);
            done += take; curPos += take;
            if (curPos >= uLen[curBlock]) { curBlock++; curPos = 0; }
        }
        return done;
    }
};

// A decompressor child (bzip2 has no in-process library in this image) started with an argument vector — the
// file name never meets a shell — whose exit status is part of the read: a failed child is an error.
struct SpawnImpl : ByteSource::Impl {
    int fd = -1;
    pid_t pid = -1;
    std::string path, tool;
    bool finished = false;
    SpawnImpl(const char *prog, const std::string &p) : path(p), tool(prog) {
        int pfd[2];
        if (pipe(pfd) != 0) throw std::runtime_error(std::string("pipe: ") + std::strerror(errno));
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_adddup2(&fa, pfd[1], 1);
        posix_spawn_file_actions_addclose(&fa, pfd[0]);
        posix_spawn_file_actions_addclose(&fa, pfd[1]);
        std::string a0 = prog, a1 = "-dc", a2 = "--", a3 = p;
        char *argv[] = {&a0[0], &a1[0], &a2[0], &a3[0], nullptr};
        const int rc = posix_spawnp(&pid, prog, &fa, nullptr, argv, environ);
        posix_spawn_file_actions_destroy(&fa);
        close(pfd[1]);
        if (rc != 0) { close(pfd[0]); pid = -1; throw std::runtime_error("Error: could not start " + tool + " for \"" + p + "\": " + std::strerror(rc)); }
        fd = pfd[0];
    }
    ~SpawnImpl() override {
        if (fd >= 0) close(fd);
        if (pid > 0 && !finished) { int st; (void)waitpid(pid, &st, 0); }      // the closed pipe ends the child
    }
    size_t read(char *dst, size_t n) override {
        size_t done = 0;
        while (done < n && !finished) {
            const ssize_t got = ::read(fd, dst + done, n - done);
            if (got < 0) { if (errno == EINTR) continue; throw std::runtime_error("Error: I/O error while reading from " + tool); }
            if (got == 0) {
                int st = 0;
                while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
                finished = true;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) cannotOpen(path);   // missing / corrupt / truncated input
                break;
            }
            done += (size_t)got;
        }
        return done;
    }
};

}  // namespace

ByteSource::ByteSource(const std::string &path, int threads) {
    if (path == "-") { impl_.reset(new PlainImpl(stdin, false, path)); return; }
    if (endsWith(path, ".bz2")) {
        std::FILE *probe = std::fopen(path.c_str(), "rb");
        if (!probe) cannotOpen(path);
        std::fclose(probe);
        impl_.reset(new SpawnImpl("bzip2", path));
        return;
    }
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) cannotOpen(path);
    if (endsWith(path, ".gz")) {
        unsigned char h[16];
        const size_t got = std::fread(h, 1, 16, f);
        std::rewind(f);
        const bool bgzf = got == 16 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
        if (bgzf) impl_.reset(new BgzfImpl(f, path, threads));
        else impl_.reset(new GzImpl(f, path));
        return;
    }
    impl_.reset(new PlainImpl(f, true, path));
}

ByteSource::~ByteSource() = default;

size_t ByteSource::read(char *dst, size_t n) { return impl_->read(dst, n); }

bool ByteSource::regularFile(int &fd, uint64_t &size) const {
    const int d = impl_->fd();
    if (d < 0) return false;
    struct stat st;
    if (fstat(d, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    fd = d; size = (uint64_t)st.st_size;
    return true;
}

}  // namespace cfamd
