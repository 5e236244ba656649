/* JSON front-end of the launcher, same interface as /root/reference/include/read_data.h:7-10 (implemented in
 * gliclass/c_amd/host/read_data.c on the host layer's own JSON reader instead of cJSON). */
#ifndef READ_DATA_H
#define READ_DATA_H

#include <stdbool.h>
#include <stddef.h>

char* read_file(const char* filename);
void parse_json(const char* json_string, char*** texts, size_t* num_texts, char**** labels, size_t** num_labels,
                size_t* num_labels_size, bool* same_labels, char** classification_type);
bool string_to_bool(const char* str);

/* Extension: `prompt_first` of the model's config.json, what /root/reference/run_GLiClass.sh:84-89 reads with jq before it starts the
 * executable.  model_path = the model directory, or a file inside it.  Returns 1 / 0, or -1 (message on stderr) if the file or a
 * boolean `prompt_first` is missing. */
int glc_config_prompt_first(const char* model_path);

/* Extension: releases what parse_json allocated (the reference never frees it, main.c:173-188). */
void free_parsed_data(char** texts, size_t num_texts, char*** labels, size_t* num_labels, bool same_labels,
                      char* classification_type);
#endif
