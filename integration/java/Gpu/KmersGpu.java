// Gpu/KmersGpu.java -- Panama FFM (JDK 22+) binding of include/kmers_gpu.h for the reference's three seams
// (INTEGRATION.md).  SOURCE ONLY: this image has no JDK, so the file has never been compiled; it is the stub a
// maintainer would start from, next to GpuFileCreator.java and GpuDB.java.  Drop the three files into src/Gpu/.
package Gpu;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_BYTE;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

/** One context = one GPU stream + its device-memory pool (kmg_create / kmg_destroy).  Internally serialised, so the
 *  reference's thread pools may share one. */
public final class KmersGpu implements AutoCloseable
{
    public static final int PAYLOAD_TAXON = 0, PAYLOAD_READPOS = 1;      // kmers_gpu.h
    public static final int EINVAL = -1, EUNSUPPORTED = -4;

    private static final Linker L = Linker.nativeLinker();
    private static final SymbolLookup LIB = SymbolLookup.libraryLookup("libkmers_gpu.so", Arena.global());

    private static MethodHandle fn(String name, FunctionDescriptor d)
    {
        return L.downcallHandle(LIB.find(name).orElseThrow(), d);
    }

    private static final MethodHandle CREATE = fn("kmg_create", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS));
    private static final MethodHandle DESTROY = fn("kmg_destroy", FunctionDescriptor.ofVoid(ADDRESS));
    private static final MethodHandle ERR = fn("kmg_last_error", FunctionDescriptor.of(ADDRESS, ADDRESS));
    private static final MethodHandle BEGIN = fn("kmg_build_begin", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT));
    private static final MethodHandle FILTERS = fn("kmg_build_set_filters", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT));
    private static final MethodHandle MIN_K = fn("kmg_build_set_min_k", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT));
    private static final MethodHandle ADD = fn("kmg_build_add", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
    private static final MethodHandle FINISH = fn("kmg_build_finish", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS));
    private static final MethodHandle DB_INFO = fn("kmg_db_info", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
    private static final MethodHandle DB_FREE = fn("kmg_db_free", FunctionDescriptor.ofVoid(ADDRESS));
    private static final MethodHandle DB_LOAD = fn("kmg_db_load", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
    private static final MethodHandle DB_LOAD_WIDE = fn("kmg_db_load_wide", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
    private static final MethodHandle ENCODE = fn("kmg_db_encode", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS));
    private static final MethodHandle MATCH = fn("kmg_match", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
    private static final MethodHandle TAXA = fn("kmg_taxacounts", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));

    final MemorySegment ctx;

    public KmersGpu(int device) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment out = a.allocate(ADDRESS);
            int rc;
            try { rc = (int) CREATE.invokeExact(device, out); } catch (Throwable t) { throw new RuntimeException(t); }
            if (rc != 0) throw new java.io.IOException("kmg_create failed (" + rc + "): is a GPU visible?");
            ctx = out.get(ADDRESS, 0);
        }
    }

    /** Nonzero return codes in the reference's own conventions (FileCreator.java:72-75, DB.java:38-41,59-63,78-81). */
    void check(int rc) throws Exception
    {
        if (rc == 0) return;
        String msg;
        try { msg = ((MemorySegment) ERR.invokeExact(ctx)).reinterpret(4096).getString(0); } catch (Throwable t) { msg = "error " + rc; }
        if (rc == EINVAL) throw new Exceptions.InconsistentDataException(msg);
        if (rc == EUNSUPPORTED) throw new UnsupportedOperationException(msg);      // the caller falls back to the Java path
        throw new java.io.IOException(msg);
    }

    private int call(MethodHandle h, Object... args)
    {
        try { return (int) h.invokeWithArguments(args); } catch (Throwable t) { throw new RuntimeException(t); }
    }

    // ---- build (FileCreator's seam) -------------------------------------------------------------------------------
    public void buildBegin(int maxK, int minK, int keyLength, boolean rc, int payloadKind, int dust, int runOfSame) throws Exception
    {
        check(call(BEGIN, ctx, maxK, keyLength, rc ? 1 : 0, payloadKind));
        if (dust >= 0 || runOfSame >= 0) check(call(FILTERS, ctx, dust, runOfSame));
        if (minK != maxK) check(call(MIN_K, ctx, minK));
    }

    /** One batch of sequences: the characters > 32 of every KMER-state run back to back, their offsets, and per sequence
     *  the taxon (reference) or a header token (reads; equal tokens iff equal header strings).
     *  @return for reads the read index of every sequence (-1: it handed out no k-mer); for taxon builds with wantFlags
     *  0 / -1 per sequence (it produced a k-mer / it did not); else null */
    public int[] buildAdd(byte[] bases, long[] seqOff, int[] taxa, long[] tokens, boolean wantFlags) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            long nseq = seqOff.length - 1;
            boolean out = tokens != null || wantFlags;
            MemorySegment b = a.allocateFrom(JAVA_BYTE, bases), o = a.allocateFrom(JAVA_LONG, seqOff);
            MemorySegment id = taxa == null ? MemorySegment.NULL : a.allocateFrom(JAVA_INT, taxa);
            MemorySegment tk = tokens == null ? MemorySegment.NULL : a.allocateFrom(JAVA_LONG, tokens);
            MemorySegment ri = out ? a.allocate(JAVA_INT, Math.max(nseq, 1)) : MemorySegment.NULL;
            check(call(ADD, ctx, b, (long) bases.length, o, id, tk, nseq, ri));
            return out ? ri.asSlice(0, 4 * nseq).toArray(JAVA_INT) : null;
        }
    }

    /** @return the database handle (kmg_db*); free with dbFree */
    public MemorySegment buildFinish() throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment db = a.allocate(ADDRESS), nwin = a.allocate(JAVA_LONG);
            check(call(FINISH, ctx, db, nwin));
            return db.get(ADDRESS, 0);
        }
    }

    /** {distinct k-mers, list entries} of a database (kmg_db_info) */
    public long[] dbInfo(MemorySegment db) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment u = a.allocate(JAVA_LONG), v = a.allocate(JAVA_LONG);
            check(call(DB_INFO, db, MemorySegment.NULL, MemorySegment.NULL, MemorySegment.NULL, MemorySegment.NULL, u, v));
            return new long[]{u.get(JAVA_LONG, 0), v.get(JAVA_LONG, 0)};
        }
    }

    public void dbFree(MemorySegment db)
    {
        try { DB_FREE.invokeExact(db); } catch (Throwable t) { throw new RuntimeException(t); }
    }

    /** Every key's block as KmerFile stores it (FileCreator.java:312-347), back to back: kmg_db_encode, two calls. */
    public record Blocks(byte[] bytes, long[] blockOff) {}

    public Blocks dbEncode(MemorySegment db, int keyLength) throws Exception
    {
        int nkeys = 1 << (2 * keyLength);
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment off = a.allocate(JAVA_LONG, nkeys + 1), total = a.allocate(JAVA_LONG);
            check(call(ENCODE, ctx, db, MemorySegment.NULL, 0L, total, off));             // sizes first
            long n = total.get(JAVA_LONG, 0);
            MemorySegment out = a.allocate(Math.max(n, 1));
            check(call(ENCODE, ctx, db, out, n, total, off));
            return new Blocks(out.asSlice(0, n).toArray(JAVA_BYTE), off.toArray(JAVA_LONG));
        }
    }

    // ---- match / counts (DB's and TaxaCounts' seams) ---------------------------------------------------------------
    public MemorySegment dbLoad(int k, int keyLength, boolean rc, int kind, long[] keys, long[] valOff, int[] valA, int[] valB) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment out = a.allocate(ADDRESS);
            check(call(DB_LOAD, ctx, k, keyLength, rc ? 1 : 0, kind, a.allocateFrom(JAVA_LONG, keys), (long) keys.length,
                       a.allocateFrom(JAVA_LONG, valOff), a.allocateFrom(JAVA_INT, valA), a.allocateFrom(JAVA_INT, valB), out));
            return out.get(ADDRESS, 0);
        }
    }

    /** dbLoad for k-mers of 33 .. 64 bases: keys = bases 0 .. 31, keysLo = bases 32 .. k-1 (kmg_db_load_wide) */
    public MemorySegment dbLoadWide(int k, int keyLength, boolean rc, int kind, long[] keys, long[] keysLo, long[] valOff, int[] valA, int[] valB) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment out = a.allocate(ADDRESS);
            // (k-mers of one length: min_k = k, no length array)
            check(call(DB_LOAD_WIDE, ctx, k, k, keyLength, rc ? 1 : 0, kind, a.allocateFrom(JAVA_LONG, keys), a.allocateFrom(JAVA_LONG, keysLo),
                       MemorySegment.NULL, (long) keys.length, a.allocateFrom(JAVA_LONG, valOff), a.allocateFrom(JAVA_INT, valA),
                       a.allocateFrom(JAVA_INT, valB), out));
            return out.get(ADDRESS, 0);
        }
    }

    /** hit[i] = index of search k-mer i among ref's k-mers, -1 = no match (DB.quickMatchCommonKey's inner join) */
    public long[] match(MemorySegment search, MemorySegment ref, long nSearch) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment hit = a.allocate(JAVA_LONG, Math.max(nSearch, 1)), nh = a.allocate(JAVA_LONG);
            check(call(MATCH, ctx, search, ref, hit, nh));
            return hit.asSlice(0, 8 * nSearch).toArray(JAVA_LONG);
        }
    }

    public record TaxaTable(int[] taxon, long[] match, long[] unique) {}

    public TaxaTable taxaCounts(MemorySegment db, int capacity) throws Exception
    {
        try (Arena a = Arena.ofConfined())
        {
            MemorySegment t = a.allocate(JAVA_INT, capacity), m = a.allocate(JAVA_LONG, capacity), u = a.allocate(JAVA_LONG, capacity);
            MemorySegment n = a.allocate(JAVA_LONG);
            n.set(JAVA_LONG, 0, capacity);
            check(call(TAXA, ctx, db, t, m, u, n));
            long used = n.get(JAVA_LONG, 0);
            return new TaxaTable(t.asSlice(0, 4 * used).toArray(JAVA_INT), m.asSlice(0, 8 * used).toArray(JAVA_LONG),
                                 u.asSlice(0, 8 * used).toArray(JAVA_LONG));
        }
    }

    @Override
    public void close()
    {
        try { DESTROY.invokeExact(ctx); } catch (Throwable t) { throw new RuntimeException(t); }
    }
}
