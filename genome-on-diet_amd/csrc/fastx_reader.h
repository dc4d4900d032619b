// Input half of SURVEY 8(f) rank 2: FASTA / FASTQ (plain or gzip) -> mini-batches, with the record grammar of the reference's
// kseq_read (LR/kseq.h:191-232) and the batching rule of mm_bseq_read3 (LR/bseq.c:80-121), so that a batch holds exactly the
// reads, names, comments and quality strings the reference would hand to its worker threads.  Plain C++ + zlib; no HIP.
//
// What differs from the reference is the mechanics: one large buffer refilled by gzread, lines located with memchr, all strings
// of a batch parsed straight into one arena (no allocation per read), plain files read without zlib's extra copy.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

struct GdFastx {
	gzFile fp = nullptr; // gzip input, or stdin
	int fd = -1;         // plain file: read() straight into the buffer (gzread would copy it once more)
	std::vector<unsigned char> buf;
	size_t begin = 0, end = 0;
	bool eof = false, io_error = false;
	int last_char = 0; // kseq_t::last_char: the header character of the next record has been consumed already
	// the record read ahead by the fragment-mode pairing rule (mm_bseq_file_t::s): name, comment, seq, qual
	bool have_pending = false;
	std::string pending[4];
	// the batch being built / handed out last: every string in `arena` (NUL-terminated), 4 offsets per read (name, comment, seq,
	// qual; -1 = absent).  Records are parsed straight into the arena.
	std::string arena, scratch;
	std::vector<int64_t> off;
	std::vector<const char *> v_name, v_comment, v_seq, v_qual;
	std::vector<int32_t> v_len;

	bool fill()
	{
		if (eof) return false;
		begin = 0;
		long n;
		if (fd >= 0) {
			do n = (long)::read(fd, buf.data(), buf.size()); while (n < 0 && errno == EINTR);
		} else n = gzread(fp, buf.data(), (unsigned)buf.size());
		if (n <= 0) { eof = true, end = 0; if (n < 0) io_error = true; return false; }
		end = (size_t)n;
		return true;
	}
	int getc_()
	{
		if (begin >= end && !fill()) return -1;
		return buf[begin++];
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
				const void *p = memchr(buf.data() + begin, '\n', end - begin);
				i = p ? (size_t)((const unsigned char *)p - buf.data()) : end;
			} else {
				for (i = begin; i < end; ++i) {
					const unsigned char c = buf[i];
					if (c == ' ' || (c >= '\t' && c <= '\r')) break; // isspace() in the C locale
				}
			}
			gotany = true;
			s.append((const char *)buf.data() + begin, i - begin);
			begin = i + 1;
			if (i < end) { if (dret) *dret = buf[i]; break; }
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
	static size_t qname_len(const char *s) // mm_qname_len (LR/bseq.h:30-36)
	{
		const size_t l = strlen(s);
		return l >= 3 && s[l - 1] >= '0' && s[l - 1] <= '9' && s[l - 2] == '/' ? l - 2 : l;
	}
	void keep(const int64_t o[4], long l_seq)
	{
		off.insert(off.end(), o, o + 4);
		v_len.push_back((int32_t)l_seq);
	}
	// mm_bseq_read3 (LR/bseq.c:80-121); returns the number of reads of the batch (0 at the end of the file), < 0 on a read error
	int read_batch(int64_t chunk_size, bool with_qual, bool with_comment, bool frag_mode, bool *parse_error)
	{
		arena.clear(), off.clear(), v_len.clear();
		if (parse_error) *parse_error = false;
		int64_t size = 0, o[4];
		long ret = 0;
		if (have_pending) { // (read with the flags of the call that looked ahead, as in the reference)
			for (int k = 0; k < 4; ++k) {
				const bool present = k == 0 || k == 2 || !pending[k].empty();
				o[k] = present ? (int64_t)arena.size() : -1;
				if (present) arena.append(pending[k]), arena.push_back('\0');
			}
			keep(o, (long)pending[2].size());
			size = (int64_t)pending[2].size(), have_pending = false;
		}
		while ((ret = read_record(with_qual, with_comment, o)) >= 0) {
			keep(o, ret);
			size += ret;
			if (size >= chunk_size) {
				if (frag_mode && ret < 1000000) { // CHECK_PAIR_THRES: keep the mates of the last read in this batch
					while ((ret = read_record(with_qual, with_comment, o)) >= 0) {
						const char *prev = arena.data() + off[off.size() - 4], *cur = arena.data() + o[0];
						const size_t l1 = qname_len(cur), l2 = qname_len(prev);
						if (l1 == l2 && strncmp(cur, prev, l1) == 0) keep(o, ret);
						else { // belongs to the next batch: take it out of this arena again
							for (int k = 0; k < 4; ++k) pending[k] = o[k] < 0 ? std::string() : std::string(arena.data() + o[k]);
							arena.resize((size_t)o[0]);
							have_pending = true;
							break;
						}
					}
				}
				break;
			}
		}
		if (ret < -1 && parse_error) *parse_error = true; // the reference warns and goes on with what it has
		if (io_error) return -1;
		const size_t n = v_len.size();
		v_name.resize(n), v_comment.resize(n), v_seq.resize(n), v_qual.resize(n);
		for (size_t i = 0; i < n; ++i) {
			v_name[i] = arena.data() + off[4 * i];
			v_comment[i] = off[4 * i + 1] < 0 ? nullptr : arena.data() + off[4 * i + 1];
			v_seq[i] = arena.data() + off[4 * i + 2];
			v_qual[i] = off[4 * i + 3] < 0 ? nullptr : arena.data() + off[4 * i + 3];
		}
		return (int)n;
	}
};

static inline GdFastx *gd_fastx_open(const char *path)
{
	GdFastx *fx = new GdFastx();
	fx->buf.resize(4 << 20);
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
	if (fx->fp) gzclose(fx->fp);
	if (fx->fd >= 0) ::close(fx->fd);
	delete fx;
}
