// cf_bytesource.hpp — the bytes of one read file, whatever its container: plain, stdin ("-"), gzip (inflated in
// this process with zlib; BGZF files block-parallel) or bzip2 (a `bzip2 -dc` child started WITHOUT a shell).
// The reference leaves decompression to its Perl wrapper (centrifuge:412-419: gzip -dc / bzip2 -dc into named
// pipes); this front end takes the compressed file directly.  A missing, truncated or corrupt input is an error
// (exception), never a silently shorter read set.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>

namespace cfamd {

class ByteSource {
public:
    // threads: helper threads a BGZF input may use for inflating blocks (>= 1)
    explicit ByteSource(const std::string &path, int threads = 1);
    ~ByteSource();
    ByteSource(const ByteSource &) = delete;
    ByteSource &operator=(const ByteSource &) = delete;
    // Fills dst with up to n bytes; returns fewer than n only at the end of the input.
    size_t read(char *dst, size_t n);
    // A plain regular file: its descriptor and size, so that a caller may pread() ranges of it from several threads
    // (read() must then not be used any more); false for stdin, pipes and compressed inputs.
    bool regularFile(int &fd, uint64_t &size) const;

    struct Impl;

private:
    std::unique_ptr<Impl> impl_;
};

}  // namespace cfamd
