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
        this.gpu = gpu;
        this.k = k;
        this.keyLength = keyLength;
        this.valOff = valOff;
        this.taxon = taxon;
        this.count = count;
        ref = gpu.dbLoad(k, keyLength, true, KmersGpu.PAYLOAD_TAXON, keys, valOff, taxon, count);
    }

    /** One batch of search k-mers (ascending, distinct, each with its read positions as CSR) -> for every one that the
     *  database holds, its index there: the caller formats the line with the unchanged ResultsDataType.toString from
     *  (search k-mer, its ReadPos set, KmerDiff "-", counts(hit)). */
    public long[] getNearestKmers(long[] searchKeys, long[] searchValOff, int[] read, int[] pos) throws Exception
    {
        MemorySegment search = gpu.dbLoad(k, keyLength, false, KmersGpu.PAYLOAD_READPOS, searchKeys, searchValOff, read, pos);
        try
        {
            return gpu.match(search, ref, searchKeys.length);
        }
        finally
        {
            gpu.dbFree(search);
        }
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
