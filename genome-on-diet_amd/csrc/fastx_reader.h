// Input half of SURVEY 8(f) rank 2: FASTA / FASTQ (plain or gzip) -> mini-batches, with the record grammar of the reference's
// kseq_read (LR/kseq.h:191-232) and the batching rule of mm_bseq_read3 (LR/bseq.c:80-121), so that a batch holds exactly the
// reads, names, comments and quality strings the reference would hand to its worker threads.  Plain C++ + zlib; no HIP.
//
// What differs from the reference is the mechanics: one large buffer refilled by gzread, lines located with memchr, all strings
// of a batch parsed straight into one arena (no allocation per read), plain files read without zlib's extra copy.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <zlib.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

// Record parser over a byte range held in memory (kseq_read's grammar); the reader below feeds it blocks of the file.
struct GdFastxParser {
	const unsigned char *b = nullptr;
	size_t begin = 0, end = 0;
	bool eof = false;  // ran off the end of the range (kseq's end-of-file behaviour applies from then on)
	int last_char = 0; // kseq_t::last_char: the header character of the next record has been consumed already
	std::string arena, scratch; // every string of the parsed records, NUL-terminated
	bool fill() { eof = true; return false; }
	int getc_()
	{
		if (begin >= end && !fill()) return -1;
		return b[begin++];
	}
	// ks_getuntil2 (LR/kseq.h:102-149) for KS_SEP_LINE (line = true) and KS_SEP_SPACE, appending to s, whose current field starts
	// at `base`; returns the length of the field, or -1 at the end of the file with nothing read
	long getuntil(bool line, std::string &s, size_t base, int *dret)
	{
		bool gotany = false;
		if (dret) *dret = 0;
		for (;;) {
			if (begin >= end && !fill()) break;
			size_t i;
			if (line) {
				const void *p = memchr(b + begin, '\n', end - begin);
				i = p ? (size_t)((const unsigned char *)p - b) : end;
			} else {
				for (i = begin; i < end; ++i) {
					const unsigned char c = b[i];
					if (c == ' ' || (c >= '\t' && c <= '\r')) break; // isspace() in the C locale
				}
			}
			gotany = true;
			s.append((const char *)b + begin, i - begin);
			begin = i + 1;
			if (i < end) { if (dret) *dret = b[i]; break; }
		}
		if (!gotany && eof) return -1;
		if (line && s.size() - base > 1 && s.back() == '\r') s.pop_back();
		return (long)(s.size() - base);
	}
	// kseq_read (LR/kseq.h:191-232) into the arena: >= 0 sequence length, -1 end of file, -2 truncated quality string.  On success
	// o[0..3] are the offsets of name, comment, seq, qual (-1: none kept); on failure the arena is as before.
	long read_record(bool with_qual, bool with_comment, int64_t o[4])
	{
		int c;
		if (last_char == 0) { // jump to the next header line
			while ((c = getc_()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return -1;
			last_char = c;
		}
		const size_t start = arena.size();
		o[0] = (int64_t)start, o[1] = o[3] = -1;
		if (getuntil(false, arena, start, &c) < 0) { arena.resize(start); return -1; }
		arena.push_back('\0');
		if (c != '\n') { // FASTA/Q comment: the rest of the header line
			if (with_comment) {
				const size_t cs = arena.size();
				if (getuntil(true, arena, cs, nullptr) > 0) o[1] = (int64_t)cs, arena.push_back('\0');
				else arena.resize(cs);
			} else scratch.clear(), getuntil(true, scratch, 0, nullptr);
		}
		const size_t ss = arena.size();
		o[2] = (int64_t)ss;
		while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue; // skip empty lines
			arena.push_back((char)c);
			getuntil(true, arena, ss, nullptr); // the rest of the line
		}
		if (c == '>' || c == '@') last_char = c;
		const size_t l_seq = arena.size() - ss;
		for (size_t i = ss; i < arena.size(); ++i) if (arena[i] == 'u' || arena[i] == 'U') --arena[i]; // U -> T (kseq2bseq, LR/bseq.c:71-73)
		arena.push_back('\0');
		if (c != '+') return (long)l_seq; // FASTA
		while ((c = getc_()) != -1 && c != '\n') {} // the rest of the '+' line
		if (c == -1) { arena.resize(start); return -2; }
		const size_t qs = arena.size();
		while (getuntil(true, arena, qs, nullptr) >= 0 && arena.size() - qs < l_seq) {}
		last_char = 0;
		if (arena.size() - qs != l_seq) { arena.resize(start); return -2; }
		if (with_qual && l_seq) o[3] = (int64_t)qs, arena.push_back('\0');
		else arena.resize(qs);
		return (long)l_seq;
	}
};

// The parsed records of one stretch of the file, in file order
struct GdFastxChunk {
	std::string arena;
	std::vector<int64_t> off;     // 4 per record: name, comment, seq, qual (-1 = none)
	std::vector<int32_t> len;     // sequence length per record
	std::vector<uint8_t> err;     // per record: a malformed record (kseq_read < -1) FOLLOWED this one ...
	bool err_first = false;       // ... or preceded the first one
	size_t next = 0;              // records handed out so far
};

struct GdFastx {
	gzFile fp = nullptr; // gzip input, or stdin
	int fd = -1;         // plain file: read() straight into the block (gzread would copy it once more)
	bool file_end = false, io_error = false;
	int n_threads = 1;
	size_t block_size = (size_t)8 << 20; // per parser thread
	// The block being parsed is bp[0, fill) = the unparsed tail of the previous block + fresh bytes; bp points into block.  While it
	// is parsed, an I/O thread reads the next stretch of the file into `other` (behind room for the tail, which is copied in front
	// of it once known), so reading and parsing overlap and no block is ever moved.
	std::vector<unsigned char> block, other;
	const unsigned char *bp = nullptr;
	size_t fill = 0;
	std::thread io;
	bool io_pending = false;
	size_t io_base = 0, io_tail = 0, io_got = 0; // other[io_base - io_tail, io_base) = tail, other[io_base, io_base + io_got) = fresh
	bool io_eof = false, io_err = false;
	std::deque<std::shared_ptr<GdFastxChunk>> ready; // parsed, not yet (completely) handed out
	std::vector<std::shared_ptr<GdFastxChunk>> lent; // every chunk the last batch points into: alive until the next call (or, detached, longer)
	std::vector<std::shared_ptr<GdFastxChunk>> pool; // used chunks, kept for their memory (fresh pages cost several times the parsing)
	std::mutex pool_mu;
	bool with_qual = true, with_comment = false, flags_set = false;
	std::vector<const char *> v_name, v_comment, v_seq, v_qual;
	std::vector<int32_t> v_len;

	static size_t qname_len(const char *s) // mm_qname_len (LR/bseq.h:30-36)
	{
		const size_t l = strlen(s);
		return l >= 3 && s[l - 1] >= '0' && s[l - 1] <= '9' && s[l - 2] == '/' ? l - 2 : l;
	}
	// read up to `want` bytes to dst; sets eof / err
	void io_read(unsigned char *dst, size_t want, size_t *got_, bool *eof_, bool *err_)
	{
		size_t got = 0;
		*eof_ = *err_ = false;
		while (got < want) {
			long n;
			if (fd >= 0) {
				do n = (long)::read(fd, dst + got, want - got); while (n < 0 && errno == EINTR);
			} else n = gzread(fp, dst + got, (unsigned)std::min<size_t>(want - got, (size_t)1 << 30));
			if (n < 0) { *err_ = true, *eof_ = true; break; }
			if (n == 0) { *eof_ = true; break; }
			got += (size_t)n;
		}
		*got_ = got;
	}
	// start reading the next `want` bytes into `other`, in front of which the tail bp[pos, fill) of the current block is placed
	void prefetch(size_t pos, size_t want)
	{
		const size_t t = fill - pos;
		io_tail = t, io_base = (t + 4095) & ~(size_t)4095;
		if (other.size() < io_base + want) other.resize(io_base + want);
		if (t) memcpy(other.data() + io_base - t, bp + pos, t);
		if (file_end) { io_got = 0, io_eof = true, io_err = false, io_pending = true; return; } // nothing more to read: only the tail moves
		unsigned char *dst = other.data() + io_base;
		io = std::thread([this, dst, want] { io_read(dst, want, &io_got, &io_eof, &io_err); });
		io_pending = true;
	}
	// make the prefetched stretch the current block
	void take_prefetched()
	{
		if (io.joinable()) io.join();
		io_pending = false;
		block.swap(other);
		bp = block.data() + io_base - io_tail;
		fill = io_tail + io_got;
		if (io_eof) file_end = true;
		if (io_err) io_error = true;
	}
	// Parse block[from, fill) sequentially into C, never starting a record at or after `stop` (the next parser's first record, or
	// `fill`).  Returns where the next record would start: exactly `stop` when the stretch ends on the seam (or the range is
	// exhausted); something else tells the caller that the seam was not a record boundary.  A record cut off by the end of the
	// block (more of the file to come) is not parsed: its start is returned.
	size_t parse_range(GdFastxChunk &C, size_t from, size_t stop, bool last_of_file) const
	{
		GdFastxParser P;
		P.b = bp, P.end = fill;
		size_t pos = from;
		P.arena.swap(C.arena);
		for (;;) {
			// the header scan of kseq_read: the next '>' or '@', wherever it stands
			size_t h = pos;
			while (h < fill && bp[h] != '>' && bp[h] != '@') ++h;
			if (h >= stop) { pos = h >= fill ? fill : h; break; } // (h == stop: the seam; h > stop: not a boundary, the caller sees it)
			P.begin = h, P.eof = false, P.last_char = 0;
			int64_t o[4];
			const long r = P.read_record(with_qual, with_comment, o);
			if (P.eof && !last_of_file) { pos = h; P.arena.resize(o[0] >= 0 && (size_t)o[0] <= P.arena.size() && r >= 0 ? (size_t)o[0] : P.arena.size()); break; } // cut off by the block end
			size_t e = P.begin;
			if (P.last_char) --e; // the next header character was consumed: hand it back
			if (r >= 0) C.off.insert(C.off.end(), o, o + 4), C.len.push_back((int32_t)r), C.err.push_back(0);
			else if (r < -1) { if (C.len.empty()) C.err_first = true; else C.err.back() = 1; }
			pos = e;
			if (r == -1 || e >= fill) { pos = std::min(e, fill); if (r == -1) pos = fill; break; }
		}
		P.arena.swap(C.arena);
		return pos;
	}
	// a position in (from, fill) that looks like the first byte of a four-line FASTQ record
	size_t find_seam(size_t from) const
	{
		const unsigned char *b = bp;
		for (size_t p = from; p < fill;) {
			const void *nl = memchr(b + p, '\n', fill - p);
			if (!nl) return fill;
			size_t q = (size_t)((const unsigned char *)nl - b) + 1; // start of a line
			if (q >= fill) return fill;
			if (b[q] == '@') {
				const void *e1 = memchr(b + q, '\n', fill - q);
				if (!e1) return fill;
				const size_t l2 = (size_t)((const unsigned char *)e1 - b) + 1;
				const void *e2 = l2 < fill ? memchr(b + l2, '\n', fill - l2) : nullptr;
				if (!e2) return fill;
				const size_t l3 = (size_t)((const unsigned char *)e2 - b) + 1;
				if (l3 < fill && b[l3] == '+') {
					const void *e3 = memchr(b + l3, '\n', fill - l3);
					if (!e3) return fill;
					const size_t l4 = (size_t)((const unsigned char *)e3 - b) + 1;
					const void *e4 = l4 < fill ? memchr(b + l4, '\n', fill - l4) : nullptr;
					if (e4 && (size_t)((const unsigned char *)e4 - b) - l4 == l3 - 1 - l2) return q;
				}
			}
			p = q;
		}
		return fill;
	}
	std::shared_ptr<GdFastxChunk> fresh_chunk()
	{
		std::shared_ptr<GdFastxChunk> c;
		{
			std::lock_guard<std::mutex> lk(pool_mu);
			if (!pool.empty()) c = std::move(pool.back()), pool.pop_back();
		}
		if (!c) c.reset(new GdFastxChunk());
		c->arena.clear(), c->off.clear(), c->len.clear(), c->err.clear(), c->err_first = false, c->next = 0;
		return c;
	}
	// read and parse one more block; false when the input is exhausted
	bool parse_more()
	{
		size_t want = block_size * (size_t)n_threads;
		for (;;) {
			if (!io_pending) { // the very first block: nothing was read ahead
				if (file_end) return false;
				if (block.size() < want) block.resize(want);
				bool e, r;
				io_read(block.data(), want, &fill, &e, &r);
				bp = block.data(), file_end = e, io_error = r;
			} else take_prefetched();
			if (fill == 0) return false;
			// split points: record starts verified by look-ahead; whether they ARE boundaries of the sequential grammar is checked
			// afterwards (every stretch must end exactly where the next one began) -- if not, the rest is parsed again in sequence
			std::vector<size_t> cut(1, 0);
			if (n_threads > 1 && fill >= std::min<size_t>((size_t)1 << 20, block_size))
				for (int k = 1; k < n_threads; ++k) {
					const size_t c = find_seam(std::max(cut.back() + 1, fill / (size_t)n_threads * (size_t)k));
					if (c >= fill) break;
					cut.push_back(c);
				}
			const size_t nr = cut.size();
			std::vector<std::shared_ptr<GdFastxChunk>> parts(nr);
			std::vector<size_t> endpos(nr, 0);
			auto work = [&](size_t k) {
				parts[k] = fresh_chunk();
				endpos[k] = parse_range(*parts[k], cut[k], k + 1 < nr ? cut[k + 1] : fill, file_end);
			};
			if (nr > 1) {
				std::vector<std::thread> th;
				for (size_t k = 1; k < nr; ++k) th.emplace_back(work, k);
				work(0);
				for (auto &t : th) t.join();
			} else work(0);
			size_t good = 1; // stretches [0, good) are what a sequential parse would have produced
			while (good < nr && endpos[good - 1] == cut[good]) ++good;
			size_t pos = endpos[good - 1];
			if (getenv("GDIET_FASTX_TRACE")) fprintf(stderr, "[fastx] block %zu bytes, %zu stretches, %zu good, file_end %d\n", fill, nr, good, (int)file_end);
			if (good < nr) { // a seam was no boundary: everything after the last good stretch again, in sequence
				parts.resize(good + 1);
				parts[good] = fresh_chunk();
				pos = parse_range(*parts[good], endpos[good - 1], fill, file_end);
			}
			size_t n_rec = 0;
			for (auto &c : parts) if (c) { n_rec += c->len.size() + (c->err_first ? 1 : 0); if (!c->len.empty() || c->err_first) ready.push_back(std::move(c)); else if (pool.size() < 64) pool.push_back(std::move(c)); }
			// the unparsed tail (a record cut off by the block end) goes in front of the next stretch of the file, which is read while
			// the caller works on what was parsed
			if (file_end && pos >= fill) { fill = 0, io_pending = false; return n_rec > 0; }
			if (!n_rec && !file_end) want *= 2; // a record longer than the block: read on, in bigger steps
			if (!n_rec && file_end) { fill = 0, io_pending = false; return false; } // (only separators / a malformed rest were left)
			prefetch(pos, want);
			if (n_rec) return true;
		}
	}
	// mm_bseq_read3 (LR/bseq.c:80-121); returns the number of reads of the batch (0 at the end of the file), < 0 on a read error
	int read_batch(int64_t chunk_size, bool wq, bool wc, bool frag_mode, bool *parse_error)
	{
		if (!flags_set) with_qual = wq, with_comment = wc, flags_set = true; // (the flags of the first call hold for the whole file)
		for (auto &c : lent) if (c.use_count() == 1 && pool.size() < 64) pool.push_back(std::move(c)); // (not in `ready` any more, not detached)
		lent.clear();
		v_name.clear(), v_comment.clear(), v_seq.clear(), v_qual.clear(), v_len.clear();
		if (parse_error) *parse_error = false;
		int64_t size = 0;
		bool closing = false; // the batch is full: only mates of its last read may still join (fragment mode)
		for (;;) {
			while (!ready.empty() && ready.front()->next >= ready.front()->len.size() && !ready.front()->err_first) ready.pop_front();
			if (ready.empty() && !parse_more()) break;
			if (ready.empty()) continue;
			GdFastxChunk &C = *ready.front();
			if (C.err_first) { C.err_first = false; if (parse_error) *parse_error = true; break; } // the reference warns and returns what it has
			if (C.next >= C.len.size()) continue;
			const size_t i = C.next;
			const char *nm = C.arena.data() + C.off[4 * i];
			if (closing) {
				const char *prev = v_name.back();
				const size_t l1 = qname_len(nm), l2 = qname_len(prev);
				if (!(l1 == l2 && strncmp(nm, prev, l1) == 0)) break;
			}
			if (lent.empty() || lent.back().get() != &C) lent.push_back(ready.front());
			v_name.push_back(nm);
			v_comment.push_back(C.off[4 * i + 1] < 0 ? nullptr : C.arena.data() + C.off[4 * i + 1]);
			v_seq.push_back(C.arena.data() + C.off[4 * i + 2]);
			v_qual.push_back(C.off[4 * i + 3] < 0 ? nullptr : C.arena.data() + C.off[4 * i + 3]);
			v_len.push_back(C.len[i]);
			size += C.len[i];
			++C.next;
			if (C.err[i]) { C.err[i] = 0; if (parse_error) *parse_error = true; break; }
			if (!closing && size >= chunk_size) {
				if (frag_mode && C.len[i] < 1000000) closing = true; // CHECK_PAIR_THRES
				else break;
			}
		}
		// (pointers into chunks still in `ready` stay valid: a chunk is only released from `lent`, at the next call)
		if (io_error) return -1;
		return (int)v_len.size();
	}
};

// a batch taken out of the reader (several mini-batches in flight): owns the pointer arrays and keeps the chunks alive
struct GdFastxBatch {
	std::vector<std::shared_ptr<GdFastxChunk>> chunks;
	std::vector<const char *> v_name, v_comment, v_seq, v_qual;
	std::vector<int32_t> v_len;
};
static inline GdFastxBatch *gd_fastx_detach(GdFastx *fx)
{
	GdFastxBatch *b = new GdFastxBatch();
	b->chunks = fx->lent; // (shared: the reader may still hand out the rest of the last chunk)
	b->v_name.swap(fx->v_name), b->v_comment.swap(fx->v_comment), b->v_seq.swap(fx->v_seq), b->v_qual.swap(fx->v_qual), b->v_len.swap(fx->v_len);
	return b;
}

static inline GdFastx *gd_fastx_open(const char *path)
{
	GdFastx *fx = new GdFastx();
	if (const char *e = getenv("GDIET_FASTX_BLOCK")) fx->block_size = std::max<size_t>(256, (size_t)atol(e)); // (tests: many small blocks)
	if (path && strcmp(path, "-")) { // a plain file is read directly, anything that starts with the gzip magic through zlib
		const int fd = ::open(path, O_RDONLY);
		if (fd < 0) { delete fx; return nullptr; }
		unsigned char magic[2] = {0, 0};
		const long n = (long)::pread(fd, magic, 2, 0);
		if (n == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
			::close(fd);
			fx->fp = gzopen(path, "r");
			if (!fx->fp) { delete fx; return nullptr; }
			gzbuffer(fx->fp, 1 << 20);
		} else fx->fd = fd;
	} else {
		fx->fp = gzdopen(0, "r"); // (LR/bseq.c:42)
		if (!fx->fp) { delete fx; return nullptr; }
	}
	return fx;
}

static inline void gd_fastx_close(GdFastx *fx)
{
	if (!fx) return;
	if (fx->io.joinable()) fx->io.join();
	if (fx->fp) gzclose(fx->fp);
	if (fx->fd >= 0) ::close(fx->fd);
	delete fx;
}
