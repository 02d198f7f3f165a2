// Gpu/GpuFileCreator.java -- FileCreator's public face (KmerFiles/FileCreator.java:33,65,117,219) over the C ABI.
// SOURCE ONLY (no JDK in this image: never compiled).  MakeDatabase constructs it instead of FileCreator when
// -K <= 64 (MakeDatabase.java:130,167), and feeds it SEQUENCES instead of a KmerStream: the record splitting
// (KmersFromFile's ID / KMER / OTHER state machine) stays in Java, windows, reverse complements, read numbering, sort,
// dedup and block encoding run on the GPU.
package Gpu;

import java.io.ByteArrayOutputStream;
import java.lang.foreign.MemorySegment;
import java.nio.ByteBuffer;
import java.util.ArrayList;
import java.util.List;

import Compression.Compressor;
import DataTypes.DataCollector;
import IndexedFiles.IndexedOutputFileSet;

public class GpuFileCreator<I, O> implements AutoCloseable
{
    private final KmersGpu gpu;
    private final int keyLength, minK, maxK;
    private final boolean rc, reads;
    private final DataCollector<I, ?, O> dataCollector;      // only its collection type id goes into the meta block

    // one batch under construction: the characters > 32 of every sequence, back to back
    private final ByteArrayOutputStream bases = new ByteArrayOutputStream(1 << 24);
    private final List<Long> seqOff = new ArrayList<>();
    private final List<Integer> taxa = new ArrayList<>();
    private final List<Long> tokens = new ArrayList<>();
    private final List<Integer> badHeader = new ArrayList<>();   // sequences of the batch whose header gave no taxon
    private static final int BATCH_BASES = 1 << 28;

    /** readIndexSink: called with (header token, read index) for every read that got an index, in order -- what
     *  ReadIDMapping.geNext prints into the -r file (ReadIDMapping.java:14-23); null for reference databases. */
    public interface ReadIndexSink { void accept(long token, int readIndex); }
    private final ReadIndexSink readIndexSink;
    private int lastIndex = -1;

    public GpuFileCreator(KmersGpu gpu, int keyLength, int minK, int maxK, DataCollector<I, ?, O> dataCollector, boolean rc,
                          boolean reads, int dust, int runOfSame, ReadIndexSink readIndexSink) throws Exception
    {
        this.gpu = gpu;
        this.keyLength = keyLength;
        this.minK = minK;
        this.maxK = maxK;
        this.rc = rc;
        this.reads = reads;
        this.dataCollector = dataCollector;
        this.readIndexSink = readIndexSink;
        seqOff.add(0L);
        gpu.buildBegin(maxK, minK, keyLength, rc, reads ? KmersGpu.PAYLOAD_READPOS : KmersGpu.PAYLOAD_TAXON, dust, runOfSame);
    }

    /** One entry into the KMER state = one sequence: its characters > 32 (KmersFromFile.java:154-158) and the header
     *  current at that time -- the taxon (Integer.parseInt(id) or the -m map) or a token of the header string.
     *  headerOk = false: the header gave no taxon (no int / not in the map).  The reference finds that out at the
     *  sequence's first k-mer and not before (KmersFromFile.java:103,165), so the sequence goes in with any taxon and
     *  flush() throws if the build reports a k-mer for it (GpuKmersFromFile is the caller). */
    public synchronized void addSequence(byte[] chars, int len, int taxon, long headerToken, boolean headerOk) throws Exception
    {
        bases.write(chars, 0, len);
        if (!reads && !headerOk) badHeader.add(seqOff.size() - 1);
        seqOff.add((long) bases.size());
        if (reads) tokens.add(headerToken); else taxa.add(taxon);
        if (bases.size() >= BATCH_BASES) flush();
    }

    private void flush() throws Exception
    {
        if (seqOff.size() == 1) return;
        long[] off = seqOff.stream().mapToLong(Long::longValue).toArray();
        int[] ids = reads ? null : taxa.stream().mapToInt(Integer::intValue).toArray();
        long[] tok = reads ? tokens.stream().mapToLong(Long::longValue).toArray() : null;
        int[] readIndex = gpu.buildAdd(bases.toByteArray(), off, ids, tok, !badHeader.isEmpty());
        for (int q : badHeader)      // taxon builds: 0 = the sequence produced a k-mer (kmg_build_add, read_index_out)
            if (readIndex[q] >= 0) throw new NumberFormatException("header of a sequence with a k-mer is no taxon");
        badHeader.clear();
        if (reads && readIndexSink != null)
            for (int s = 0; s < readIndex.length; s++)
                if (readIndex[s] >= 0 && readIndex[s] != lastIndex)
                {
                    readIndexSink.accept(tok[s], readIndex[s]);
                    lastIndex = readIndex[s];
                }
        bases.reset();
        seqOff.clear();
        seqOff.add(0L);
        taxa.clear();
        tokens.clear();
    }

    /** FileCreator.create (:117-217): the meta block, then every key's block in ascending key order through the unchanged
     *  IndexedOutputFileSet (the reference enforces that order with OrderedLatches, FileCreator.java:385-391). */
    public void create(IndexedOutputFileSet<Integer> out, boolean hr, boolean verbose) throws Exception
    {
        if (hr) throw new UnsupportedOperationException("-h: write through FileCreator");      // or format kmg_db_fetch's arrays
        flush();
        MemorySegment db = gpu.buildFinish();
        try
        {
            byte[] id = Compressor.getByteID(dataCollector.getCollectionDataType().getID());
            ByteBuffer meta = ByteBuffer.allocate(4 + id.length);
            meta.put((byte) minK).put((byte) maxK).put((byte) keyLength).put(id).put(rc ? (byte) 1 : (byte) 0);
            out.writeAll(meta.array(), -1);
            KmersGpu.Blocks blocks = gpu.dbEncode(db, keyLength);
            int maxkey = 1 << (2 * keyLength);
            for (int key = 0; key < maxkey; key++)
            {
                int from = (int) blocks.blockOff()[key], to = (int) blocks.blockOff()[key + 1];      // (files > 2 GB: encode per key range)
                out.setCurrentKey(key);
                out.write(blocks.bytes(), from, to - from);
            }
            out.close();
        }
        finally
        {
            gpu.dbFree(db);
        }
    }

    @Override
    public void close()
    {
    }
}
