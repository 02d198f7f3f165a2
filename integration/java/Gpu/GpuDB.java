// Gpu/GpuDB.java -- DB.getNearestKmers' quick branch (Database/DB.java:76-90,129-137: maxDiff = 0, search k-mers of the
// database's one length) over the C ABI.  SOURCE ONLY (no JDK in this image: never compiled).  Matcher builds it from
// the same KmerFiles it hands to DB (Matcher.java:80-101) and falls back to DB on UnsupportedOperationException
// (k > 32; several lengths go through kmg_match_trie, inexact search through kmg_match_near -- INTEGRATION.md).
package Gpu;

import java.lang.foreign.MemorySegment;
import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import java.util.TreeMap;

public class GpuDB implements AutoCloseable
{
    private final KmersGpu gpu;
    private final MemorySegment ref;
    private final int k, keyLength;
    // host copy of the reference's value lists: what a match line prints (CountDataType.toString: TxC|TxC)
    private final long[] valOff;
    private final int[] taxon, count;

    /** keys: every stored k-mer as its compressedBytes read big-endian (Sequence.cBytes, left-aligned 2 bits per base),
     *  ascending; valOff / taxon / count: the CSR of its TreeCountMap entries in ascending taxon order. */
    public GpuDB(KmersGpu gpu, int k, int keyLength, long[] keys, long[] valOff, int[] taxon, int[] count) throws Exception
    {
        this(gpu, k, keyLength, keys, null, valOff, taxon, count);
    }

    /** keysLo (k > 32 only, else null): bytes 9 .. 16 of every k-mer's compressedBytes, read the same way (bases 32 .. k-1) */
    public GpuDB(KmersGpu gpu, int k, int keyLength, long[] keys, long[] keysLo, long[] valOff, int[] taxon, int[] count) throws Exception
    {
        this.gpu = gpu;
        this.k = k;
        this.keyLength = keyLength;
        this.valOff = valOff;
        this.taxon = taxon;
        this.count = count;
        ref = k > 32 ? gpu.dbLoadWide(k, keyLength, true, KmersGpu.PAYLOAD_TAXON, keys, keysLo, valOff, taxon, count)
                     : gpu.dbLoad(k, keyLength, true, KmersGpu.PAYLOAD_TAXON, keys, valOff, taxon, count);
    }

    /** bytes first .. first + 7 of a k-mer's compressedBytes (cb[0] is the length byte), big-endian, zero-padded */
    static long word(byte[] cb, int first)
    {
        long v = 0;
        for (int b = first; b < first + 8; b++) v = (v << 8) | (b < cb.length ? (cb[b] & 0xFFL) : 0L);
        return v;
    }

    /** The reference database `db` (one or several files: DB.kmers merges them, DB.java:99-109) on the GPU, or null when it
     *  is not one the GPU path takes (k-mers of several lengths, k > 64): key by key, every entry's k-mer and its
     *  TreeCountMap in ascending taxon order.  Matcher and TaxaCounts call it once, after `new DB<>(dbfiles, merger)`. */
    public static GpuDB fromDB(Database.DB<CountMaps.TreeCountMap<Integer>> db) throws Exception
    {
        Kmers.KmerStream<CountMaps.TreeCountMap<Integer>> first = db.kmers(0);
        int minK = first.getMinLength(), maxK = first.getMaxLength();
        first.close();
        if (minK != maxK || maxK > 64) return null;
        int keyLength = Integer.numberOfTrailingZeros(db.getMaxKey()) / 2;                 // maxKey = 4^keyLength
        java.util.ArrayList<Long> keys = new java.util.ArrayList<>(), keysLo = new java.util.ArrayList<>(), off = new java.util.ArrayList<>();
        java.util.ArrayList<Integer> taxon = new java.util.ArrayList<>(), count = new java.util.ArrayList<>();
        off.add(0L);
        for (int key = 0; key < db.getMaxKey(); key++)
            db.kmers(key).forEach(kwd -> {
                byte[] cb = kwd.getKmer().compressedBytes();
                keys.add(word(cb, 1));
                keysLo.add(word(cb, 9));
                for (Map.Entry<Integer, Long> e : kwd.getData().entrySet())
                {
                    taxon.add(e.getKey());
                    count.add((int) Math.min(e.getValue(), Integer.MAX_VALUE));
                }
                off.add((long) taxon.size());
            });
        return new GpuDB(new KmersGpu(0), maxK, keyLength, keys.stream().mapToLong(Long::longValue).toArray(),
                         maxK > 32 ? keysLo.stream().mapToLong(Long::longValue).toArray() : null,
                         off.stream().mapToLong(Long::longValue).toArray(), taxon.stream().mapToInt(Integer::intValue).toArray(),
                         count.stream().mapToInt(Integer::intValue).toArray());
    }

    /** One batch of search k-mers (ascending, distinct, each with its read positions as CSR) -> for every one that the
     *  database holds, its index there: the caller formats the line with the unchanged ResultsDataType.toString from
     *  (search k-mer, its ReadPos set, KmerDiff "-", counts(hit)). */
    public long[] getNearestKmers(long[] searchKeys, long[] searchValOff, int[] read, int[] pos) throws Exception
    {
        return getNearestKmers(searchKeys, null, searchValOff, read, pos);
    }

    public long[] getNearestKmers(long[] searchKeys, long[] searchKeysLo, long[] searchValOff, int[] read, int[] pos) throws Exception
    {
        MemorySegment search = k > 32 ? gpu.dbLoadWide(k, keyLength, false, KmersGpu.PAYLOAD_READPOS, searchKeys, searchKeysLo, searchValOff, read, pos)
                                      : gpu.dbLoad(k, keyLength, false, KmersGpu.PAYLOAD_READPOS, searchKeys, searchValOff, read, pos);
        try
        {
            return gpu.match(search, ref, searchKeys.length);
        }
        finally
        {
            gpu.dbFree(search);
        }
    }

    /** DB.quickMatchCommonKey (Database/DB.java:129-137) for one key's search k-mers, as DB calls it from its pool
     *  threads (DB.java:199-201; a context is internally serialised): the k-mers of the list that the database holds,
     *  each with a ClosestInfoCollector over the database's entry, in the list's (ascending) order -- what
     *  StreamUtils.matchTwoStreams hands on.  `make` turns (taxa, counts) of an entry into the database's data type
     *  (TreeCountMap for reference databases). */
    public <S, D> List<Kmers.KmerWithData<DataTypes.DataPair<S, Database.ClosestInfoCollector<D>>>> quickMatchCommonKey(
            List<Kmers.KmerWithData<S>> kmers, java.util.function.Function<Map<Integer, Long>, D> make) throws Exception
    {
        long[] keys = new long[kmers.size()], keysLo = new long[kmers.size()];
        for (int i = 0; i < keys.length; i++)
        {
            byte[] cb = kmers.get(i).getKmer().compressedBytes();      // [length][2 bits per base, first base in the top bits]
            keys[i] = word(cb, 1);
            keysLo[i] = word(cb, 9);
        }
        long[] valOff = new long[keys.length + 1];                      // the join needs the k-mers only
        MemorySegment search = k > 32 ? gpu.dbLoadWide(k, keyLength, false, KmersGpu.PAYLOAD_READPOS, keys, keysLo, valOff, new int[0], new int[0])
                                      : gpu.dbLoad(k, keyLength, false, KmersGpu.PAYLOAD_READPOS, keys, valOff, new int[0], new int[0]);
        long[] hit;
        try { hit = gpu.match(search, ref, keys.length); }
        finally { gpu.dbFree(search); }
        List<Kmers.KmerWithData<DataTypes.DataPair<S, Database.ClosestInfoCollector<D>>>> out = new ArrayList<>();
        for (int i = 0; i < keys.length; i++)
            if (hit[i] >= 0)
            {
                Kmers.KmerWithData<S> s = kmers.get(i);
                Kmers.KmerWithData<D> entry = new Kmers.KmerWithData<>(s.getKmer(), make.apply(counts(hit[i])));
                out.add(new Kmers.KmerWithData<>(s.getKmer(), new DataTypes.DataPair<>(s.getData(), new Database.ClosestInfoCollector<>(entry))));
            }
        return out;
    }

    /** The TreeCountMap of database k-mer `index` (counts as the file stores them: DB.java:102 adds them as longs). */
    public Map<Integer, Long> counts(long index)
    {
        Map<Integer, Long> m = new TreeMap<>();
        for (long i = valOff[(int) index]; i < valOff[(int) index + 1]; i++) m.put(taxon[(int) i], (long) count[(int) i]);
        return m;
    }

    /** TaxaCounts.SingleIndexAll over the whole database (TaxaCounts.java:355-360): rows in ascending taxon order. */
    public List<long[]> taxaCounts(int maxTaxa) throws Exception
    {
        KmersGpu.TaxaTable t = gpu.taxaCounts(ref, maxTaxa);
        List<long[]> rows = new ArrayList<>();
        for (int i = 0; i < t.taxon().length; i++) rows.add(new long[]{t.taxon()[i], t.match()[i], t.unique()[i]});
        return rows;
    }

    @Override
    public void close()
    {
        gpu.dbFree(ref);
    }
}
