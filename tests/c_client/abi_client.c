/* A plain-C client of include/dsvt_plugin.h: what a maintainer of the reference's host code would write instead of the
 * getPluginRegistry()->getPluginCreator(...)->createPlugin(...) sequence of include/plugin_helper.h:253-310.
 * Runs without a GPU: creator side only (field names, createPlugin, output shapes, serialise / deserialise, clone). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dsvt_plugin.h"

static int fail(const char* what) { fprintf(stderr, "FAIL: %s\n", what); return 1; }

int main(void)
{
    int n = dsvtGetNbPluginTypes(), i, found = 0;
    if (n < 10) return fail("fewer than the reference's ten plugin types");
    for (i = 0; i < n; ++i) if (!strcmp(dsvtGetPluginTypeName(i), "GetSetPlugin")) found = 1;
    if (!found) return fail("GetSetPlugin not registered");

    /* add_get_set_op (plugin_helper.h:253-310): walk the advertised names, fill the ones we know */
    const DsvtPluginFieldCollection* adv = dsvtGetFieldNames("GetSetPlugin", "1");
    if (!adv || adv->nbFields != 4) return fail("GetSetPlugin field list");
    int max_win_num = 800, max_voxel_num_per_win = 576, voxel_num_set = 36, win_shape[3] = {12, 12, 1};
    DsvtPluginField f[4];
    for (i = 0; i < adv->nbFields; ++i) {
        const char* name = adv->fields[i].name;
        f[i].name = name; f[i].type = DSVT_FIELD_INT32; f[i].length = 1;
        if (!strcmp(name, "max_win_num")) f[i].data = &max_win_num;
        else if (!strcmp(name, "max_voxel_num_per_win")) f[i].data = &max_voxel_num_per_win;
        else if (!strcmp(name, "voxel_num_set")) f[i].data = &voxel_num_set;
        else if (!strcmp(name, "win_shape")) { f[i].data = win_shape; f[i].length = 3; }
        else return fail("unexpected field name");
    }
    DsvtPluginFieldCollection fc; fc.nbFields = 4; fc.fields = f;
    DsvtPlugin* p = dsvtCreatePlugin("GetSetPlugin", "1", "get_set_layer", &fc);
    if (!p) return fail("createPlugin");
    if (strcmp(dsvtPluginGetType(p), "GetSetPlugin") || strcmp(dsvtPluginGetVersion(p), "1")) return fail("type / version");
    if (dsvtPluginGetNbOutputs(p) != 5) return fail("GetSet has five outputs (getSet.cu:161)");

    /* inputs as WindowPartition produces them: gidx [1,800,576], cinw [1,800,576,3], vcnt [1,800], win_num [1] */
    DsvtDims in[4]; memset(in, 0, sizeof in);
    in[0].nbDims = 3; in[0].d[0] = 1; in[0].d[1] = 800; in[0].d[2] = 576;
    in[1].nbDims = 4; in[1].d[0] = 1; in[1].d[1] = 800; in[1].d[2] = 576; in[1].d[3] = 3;
    in[2].nbDims = 2; in[2].d[0] = 1; in[2].d[1] = 800;
    in[3].nbDims = 1; in[3].d[0] = 1;
    DsvtDims out;
    if (dsvtPluginGetOutputDimensions(p, 0, in, 4, &out) != 0) return fail("getOutputDimensions");
    if (out.nbDims != 4 || out.d[0] != 1 || out.d[1] != 2 || out.d[2] != 800 || out.d[3] != 36) return fail("inds shape [1,2,800,36]");

    /* serialise -> deserialise -> identical bytes (getSet.cu:744-758: six int32) */
    size_t sz = dsvtPluginGetSerializationSize(p);
    if (sz != 6 * sizeof(int32_t)) return fail("serialisation size");
    int32_t blob[6], blob2[6];
    dsvtPluginSerialize(p, blob);
    if (blob[0] != 36 || blob[1] != 800 || blob[2] != 576 || blob[3] != 12 || blob[4] != 12 || blob[5] != 1) return fail("serialisation layout");
    DsvtPlugin* q = dsvtDeserializePlugin("GetSetPlugin", "1", "get_set_layer", blob, sz);
    if (!q) return fail("deserializePlugin");
    dsvtPluginSerialize(q, blob2);
    if (memcmp(blob, blob2, sz)) return fail("round trip");
    DsvtPlugin* c = dsvtPluginClone(p);
    if (!c || dsvtPluginGetSerializationSize(c) != sz) return fail("clone");
    dsvtPluginDestroy(c); dsvtPluginDestroy(q); dsvtPluginDestroy(p);

    if (dsvtCreatePlugin("NoSuchPlugin", "1", "x", &fc) != NULL) return fail("unknown type must give NULL");
    if (dsvtGetFieldNames("GetSetPlugin", "2") != NULL) return fail("only version 1 is registered");
    printf("abi_client ok: %d plugin types, build %s\n", n, dsvtGetBuildInfo());
    return 0;
}
