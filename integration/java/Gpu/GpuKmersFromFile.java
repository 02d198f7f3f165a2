// Gpu/GpuKmersFromFile.java -- the record splitter of the build seam: KmersFromFile.streamFromFile's ID / KMER / OTHER
// machine (OtherFiles/KmersFromFile.java:84-208, transition tables OtherFiles/KmersFromFileStateChanger.java:37-68)
// with the k-mer work taken out.  Where the reference forms a window per character and hands a KmerStream to
// FileCreator.addKmers, this walks the same states over the same characters and hands whole SEQUENCES -- one per entry
// into the KMER state -- to GpuFileCreator.addSequence; windows, validity, reverse complements, read numbering and the
// suffix k-mers of minK < maxK are the library's (kmg_build_add, include/kmers_gpu.h).
// SOURCE ONLY (no JDK in this image: never compiled).  C++ twin, tested: kmers_amd/csrc/host_parse.cpp.
package Gpu;

import java.io.BufferedReader;
import java.io.IOException;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

public final class GpuKmersFromFile
{
    /** KmersFromFileStateChanger.getFAinstance / getFQinstance / getOldinstance (:37-68) */
    public enum Format { FASTA, FASTQ, OLD }

    private enum State { ID, KMER, OTHER }

    private final Format format;
    private final int minK, maxK;
    private final Map<String, Integer> idMap;            // MakeDatabase -m (MakeDatabase.java:138-147), or null: Integer.parseInt

    // reads: header string -> token (kmg_build_add wants tokens that are equal iff the header strings are; ReadIDMapping
    // compares a read's header with that of the last read that produced a k-mer, ReadIDMapping.java:14-23, so the
    // table may be cleared whenever a batch has been flushed and the carried header is entered again)
    private final Map<String, Long> tokens = new HashMap<>();
    private final List<String> names = new ArrayList<>();

    public GpuKmersFromFile(Format format, int minK, int maxK, Map<String, Integer> idMap)
    {
        this.format = format;
        this.minK = minK;
        this.maxK = maxK;
        this.idMap = idMap;
    }

    /** header string of a token handed to addSequence (for the -r read-name map: GpuFileCreator.ReadIndexSink) */
    public String name(long token)
    {
        return names.get((int) token);
    }

    /** the transition tables: the state a character leads to, or null (KmersFromFileStateChanger.get) */
    private State next(State s, int c)
    {
        switch (format)
        {
            case FASTA:                                   // :46-58
                if (s == State.ID && c == ' ') return State.OTHER;
                if (s == State.ID && c == '\n') return State.KMER;
                if (s == State.OTHER && c == '\n') return State.KMER;
                if (s == State.KMER && c == '\n') return State.OTHER;
                if (s == State.OTHER && c == '>') return State.ID;
                return null;
            case FASTQ:                                   // :60-68
                if (s == State.OTHER && c == '@') return State.ID;
                if (s == State.KMER && c == '@') return State.ID;
                if (s == State.ID && c == '\n') return State.KMER;
                if (s == State.KMER && c == '+') return State.OTHER;
                return null;
            default:                                      // :37-44
                if (s == State.ID && (c == '\t' || c == ' ')) return State.KMER;
                if (s == State.KMER && c == '\n') return State.ID;
                return null;
        }
    }

    /** Reads `in` to its end and hands every sequence to dbc.  A header is turned into its taxon only for the error
     *  message's sake here: the reference calls its mapper at a sequence's first k-mer and not before
     *  (new Kmer(..) is evaluated before mapper.apply, KmersFromFile.java:103,165), so a header that is no int -- or not in
     *  the -m map -- is passed on as "unknown" and GpuFileCreator throws only if the build reports a k-mer for it. */
    public void feed(BufferedReader in, GpuFileCreator<?, ?> dbc) throws Exception
    {
        State state = format == Format.OLD ? State.ID : State.OTHER;          // startState()
        StringBuilder id = new StringBuilder();
        byte[] seq = new byte[1 << 16];
        int len = 0;
        boolean inSeq = false, lastWasBase = false;
        int c;
        while ((c = in.read()) != -1)
        {
            State to = next(state, c);
            if (to != null)
            {
                if (state == State.KMER && inSeq)                               // leaving a sequence line: the sequence is complete
                {
                    emit(dbc, id.toString(), seq, len);
                    inSeq = false;
                }
                state = to;
                switch (state)
                {
                    case KMER: len = 0; inSeq = true; lastWasBase = false; break;          // pos = 0 (:137-139)
                    case ID: id = new StringBuilder(); break;                              // :140-143
                    default: break;
                }
            }
            else if (state == State.KMER)
            {
                if (c > 32)                                                       // every such character advances pos (:154-158)
                {
                    if (c >= 128) throw new UnsupportedOperationException("bytes >= 128 in a sequence line");   // Sequence.java:15-33
                    if (len == seq.length) seq = java.util.Arrays.copyOf(seq, 2 * len);
                    seq[len++] = (byte) c;
                    lastWasBase = true;
                }
                else lastWasBase = false;                                         // kwd = null (:180-183)
            }
            else if (state == State.ID) id.appendCodePoint(c);                    // :185-187
        }
        if (state == State.KMER && inSeq)
        {
            // The input ends inside a sequence line.  The reference then hands its last k-mer out once more (kwd survives
            // into the call that meets the end, KmersFromFile.java:115-123,193-200) and, with minK < maxK, further suffixes.
            // One k-mer length, FASTA: the last maxK characters as one more sequence under the same header give exactly
            // that extra k-mer when (and only when) the last window is a k-mer.  Anything else: the Java path.
            if (format == Format.FASTQ || minK != maxK)
                throw new UnsupportedOperationException("input ends inside a sequence line");
            emit(dbc, id.toString(), seq, len);
            if (lastWasBase && len >= maxK) emit(dbc, id.toString(), java.util.Arrays.copyOfRange(seq, len - maxK, len), maxK);
        }
    }

    private void emit(GpuFileCreator<?, ?> dbc, String header, byte[] seq, int len) throws Exception
    {
        if (format == Format.FASTQ)
        {
            // ReadIDMapping starts with curReadS = "" (ReadIDMapping.java:10-16): a first read with an empty header would get
            // index -1 there
            if (header.isEmpty()) throw new UnsupportedOperationException("a FASTQ record with an empty header");
            Long t = tokens.get(header);
            if (t == null)
            {
                t = (long) names.size();
                tokens.put(header, t);
                names.add(header);
            }
            dbc.addSequence(seq, len, 0, t, true);
            return;
        }
        Integer taxon;
        if (idMap != null) taxon = idMap.get(header);                            // KmersFromFile.java:44-47
        else
        {
            try { taxon = Integer.parseInt(header); }                           // :39-42
            catch (NumberFormatException ex) { taxon = null; }
        }
        dbc.addSequence(seq, len, taxon == null ? 0 : taxon, 0L, taxon != null);
    }
}
