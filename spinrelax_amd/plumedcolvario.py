"""
Reader of PLUMED PRINT output (colvar files), mirroring the reference's plumedcolvario.py:24-81 (same return
value, same messages, float32 entries because PLUMED writes single precision).  The orientation trajectory
`colvar-qorient` (fields `time q.w q.x q.y q.z`, README.md:193-198) is what spinrelax_amd.ct.detumble_vectors
consumes.
"""
import numpy as np


def read_from_plumedprint(fname):
    """-> [field_names (nfields), parsed_data (nfields, nentries) float32] or -1 on a malformed file."""
    bHeaderRead = False
    ncomm = nempty = nfields = 0
    ntot = -1
    field_names = []
    parsed = []
    with open(fname) as fp:
        for ntot, line in enumerate(fp):
            if line == '\n':
                nempty += 1
                continue
            if line.startswith('#'):
                ncomm += 1
                l = line.split()
                if len(l) > 1 and l[1] == 'FIELDS':
                    if bHeaderRead:
                        comp = l[2:]
                        for a, b in zip(field_names, comp):
                            if a != b:
                                print('= = ERROR: Multiple FIELD headers are present to indicate parallel trajectoreies, but their entries do not agree!')
                                print(field_names)
                                print(comp)
                                return -1
                    else:
                        field_names = l[2:]
                        nfields = len(field_names)
                        bHeaderRead = True
                continue
            if not bHeaderRead:
                print('= = ERROR: Data-like line encountered before a FIELDS definition! Line as follows:')
                print(line)
                return -1
            l = line.split()
            if len(l) != nfields:
                print('= = ERROR: Data-like line does not have the same number of fields as defined in FIELDS! ( %i )' % (nfields))
                print(l)
                return -1
            parsed.append(np.array(l, dtype=np.float32))
    ndata = ntot + 1 - ncomm - nempty
    print('= = Input file %s has been read: Found %i data-like lines in input plumed FES file, with %i comment lines. ' % (fname, ndata, ncomm))
    if nempty > 0:
        print('= = = NOTE: There are %i empty lines' % nempty)
    print('= = = %i field entries discovered. Field entries are as follows:' % nfields)
    print(str(field_names).strip('[]'))
    data = np.array(parsed, dtype=np.float32).reshape(ndata, nfields).T
    return field_names, np.asfortranarray(data)


def read_qorient(fname):
    """(time (N,), q (N, 4) float32 as w x y z) from a colvar-qorient file."""
    res = read_from_plumedprint(fname)
    if res == -1:
        raise ValueError('%s is not a PLUMED PRINT file' % fname)
    names, data = res
    try:
        cols = [names.index(k) for k in ('q.w', 'q.x', 'q.y', 'q.z')]
    except ValueError:
        raise ValueError('%s has no q.w q.x q.y q.z fields (found %s)' % (fname, names))
    t = data[names.index('time')] if 'time' in names else np.arange(data.shape[1], dtype=np.float32)
    return t, np.ascontiguousarray(data[cols].T)
