// Gpu/GpuMakeDatabase.java -- the wiring MakeDatabase.main does for -a and -q (Database/MakeDatabase.java:127-185,
// 279-318) with GpuFileCreator + GpuKmersFromFile in place of FileCreator + KmersFromFile.  One static call from
// MakeDatabase.main (integration/java/callsites.diff); UnsupportedOperationException = not on the GPU path, the caller
// carries on with the Java path.  SOURCE ONLY (no JDK in this image: never compiled).
package Gpu;

import java.io.BufferedOutputStream;
import java.io.BufferedReader;
import java.io.File;
import java.io.FileOutputStream;
import java.io.OutputStreamWriter;
import java.io.PrintWriter;
import java.util.HashMap;
import java.util.Map;
import java.util.zip.GZIPOutputStream;

import org.apache.commons.cli.CommandLine;

import Compression.IntCompressor;
import DataTypes.DataCollector;
import IndexedFiles.IndexedOutputFile;
import IndexedFiles.IndexedOutputFileSet;
import Zip.ZipOrNot;

public final class GpuMakeDatabase
{
    /** what the GPU path takes: -a / -q, -K <= 64, no -h, no -U, no temp files (MakeDatabase.java:106-111) */
    public static boolean usable(CommandLine commands, int minK, int maxK, boolean useExistingTemp)
    {
        return (commands.hasOption('a') || commands.hasOption('q')) && !commands.hasOption('O') && maxK <= 64
                && !commands.hasOption('h') && !useExistingTemp;
    }

    public static void run(CommandLine commands, int minK, int maxK, int keyLength, long maxSize) throws Exception
    {
        int dust = commands.hasOption('D') ? Integer.parseInt(commands.getOptionValue('D')) : -1;         // :281-288
        int runOfSame = commands.hasOption('R') ? Integer.parseInt(commands.getOptionValue('R')) : -1;
        boolean reads = commands.hasOption('q');
        try (KmersGpu gpu = new KmersGpu(0))
        {
            Map<String, Integer> idMap = null;
            if (!reads && commands.hasOption('m'))                                                         // :138-147
            {
                idMap = new HashMap<>();
                final Map<String, Integer> m = idMap;
                ZipOrNot.getBufferedReader(new File(commands.getOptionValue('m'))).lines().forEach(line -> {
                    String[] parts = line.split("\t");
                    m.put(parts[0], Integer.parseInt(parts[1]));
                });
            }
            GpuKmersFromFile kf = new GpuKmersFromFile(reads ? GpuKmersFromFile.Format.FASTQ : GpuKmersFromFile.Format.FASTA,
                                                      minK, maxK, idMap);
            PrintWriter outReadMap = reads ? new PrintWriter(new OutputStreamWriter(new GZIPOutputStream(new BufferedOutputStream(
                    new FileOutputStream(new File(commands.getOptionValue('r'))))))) : null;                // :172-173
            GpuFileCreator.ReadIndexSink sink = reads ? (token, index) -> outReadMap.println(kf.name(token)) : null;   // ReadIDMapping.java:14-23
            try (GpuFileCreator<?, ?> dbc = reads
                    ? new GpuFileCreator<>(gpu, keyLength, minK, maxK, DataCollector.getReadPosInstance(), false, true, dust, runOfSame, sink)
                    : new GpuFileCreator<>(gpu, keyLength, minK, maxK, DataCollector.getCountInstance(), true, false, dust, runOfSame, null))
            {
                BufferedReader in = ZipOrNot.getBufferedReader(new File(commands.getOptionValue('i')));    // :158
                kf.feed(in, dbc);
                IndexedOutputFileSet<Integer> out;                                                         // create(), :293-318
                if (commands.hasOption('Z'))
                    out = new IndexedOutputFileSet<>(f -> new IndexedOutputFile<>(f, new IntCompressor(), false, maxSize),
                                                     new File(commands.getOptionValue('o')));
                else
                    out = new IndexedOutputFileSet<>(f -> new IndexedOutputFile<>(f, new IntCompressor(), false,
                            Integer.parseInt(commands.getOptionValue('z', "5")), maxSize), new File(commands.getOptionValue('o')));
                dbc.create(out, false, commands.hasOption('v'));
            }
            finally
            {
                if (outReadMap != null) outReadMap.close();
            }
        }
    }
}
